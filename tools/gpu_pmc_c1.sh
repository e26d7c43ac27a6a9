#!/bin/bash
# SQ counters of the complex128 kernels of configuration C1 (tools/c1_prof.py)
export TMPDIR=/tmp
TAG=${1:-pmc_c1}
rm -rf gpurun_out/${TAG}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/${TAG} -- python3 tools/c1_prof.py > /dev/null 2> gpurun_out/${TAG}.err; echo "rc=$?"
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/${TAG}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = "k_time_mid<double>" if ("k_time<double" in k and ", 1, false>" in k) else ("k_freq<double>" if "k_freq<double" in k else None)
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v)/len(v)) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
find gpurun_out/${TAG} -name "*.csv" -size +1M -delete
