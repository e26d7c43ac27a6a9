#!/bin/bash
# SQ-level counters of the filter kernels (BPF 2^20 x 2 complex128, LPF 2^20 float64: tools/sos_prof.py), one counter group per pass
#   bash tools/gpu_pmc_sos.sh TAG > gpurun_out/TAG_sos_sq.txt
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
        k = next((t for t in ("k_filtfilt<2, 2, 4>", "k_filtfilt<2, 1, 4>", "k_apply<2, 2", "k_chunk_scan<2, 2") if t in k), None)      # (the one-launch kernel of BPF / LPF at 2^20; the three-launch pair)
        if k: acc[("long chunk " if "chunk_long" in r["Kernel_Name"] else "") + k + ">"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in sorted(d.items())}
    print(k, "dispatches", max(len(v) for v in d.values()), {c: round(v) for c, v in m.items()})
    if m.get("SQ_WAVE_CYCLES"):
        print("   fractions of SQ_WAVE_CYCLES:", {c: round(v / m["SQ_WAVE_CYCLES"], 3) for c, v in m.items() if c.startswith("SQ_") and c not in ("SQ_WAVE_CYCLES", "SQ_WAVES", "SQ_BUSY_CYCLES")})
PY
find gpurun_out/${TAG} -name "*.csv" -size +1M -delete
