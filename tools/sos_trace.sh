#!/bin/bash
# Per-launch durations and gaps of the last device-resident filter calls (dev aid): bash tools/sos_trace.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/sostrace
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/sostrace -- python3 $R/tools/sos_prof.py > $R/gpurun_out/sostrace_run.txt 2>&1
tail -3 $R/gpurun_out/sostrace_run.txt
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/sostrace/**/*kernel_trace.csv", recursive=True)[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "<2, 2, 4>" in r["Kernel_Name"]]
tail = rows[-30:]
prev = None
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].split("::")[1][:14]
    print(f"{nm:14s} {(e - s) / 1e3:6.1f} us  gap {((s - prev) / 1e3 if prev else 0):5.1f}  grid {r.get('Grid_Size', r.get('Grid_Size_X'))} lds {r.get('LDS_Block_Size')} vgpr {r.get('VGPR_Count')}")
    prev = e
PY
