#!/bin/bash
# SQ-level counters of the two filter kernels (BPF 2^20 x 2 complex128), one counter group per pass
export TMPDIR=/tmp
TAG=${1:-pmc_sos}
rm -rf gpurun_out/${TAG}
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS"; do
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/${TAG}/$(echo $grp | tr ' ' '_') -- python3 tools/sos_prof.py > /dev/null 2>> gpurun_out/${TAG}.err; echo "rc=$? ($grp)"
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/${TAG}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = "k_apply<2,2>" if "k_apply<2, 2>" in k else ("k_chunk_scan<2,2>" if "k_chunk_scan<2, 2>" in k else None)
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v)/len(v)) for c, v in sorted(d.items())})
PY
find gpurun_out/${TAG} -name "*.csv" -size +1M -delete
