#!/bin/bash
# SQ-level counters for the two hot kernels (single lane for clean per-kernel numbers)
export TMPDIR=/tmp
TAG=${1:-pmc2}
rm -rf gpurun_out/${TAG}
SSFM_LANES=${LANES:-1} SSFM_E=${E:-16} rocprofv3 --pmc ${CTRS:-SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE} --output-format csv -d gpurun_out/${TAG} -- python3 bench.py --steps 1 --warmup 0 --cpu-steps 0 --no-profile-pass > /dev/null 2> gpurun_out/${TAG}.err; echo "rc=$?"
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/${TAG}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = "k_time_mid" if ("k_time<float" in k and (", 1>" in k or ", 1, true>" in k or ", 1, false>" in k)) else ("k_freq" if "k_freq<float" in k else None)
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v)/len(v)) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
find gpurun_out/${TAG} -name "*.csv" -size +1M -delete
