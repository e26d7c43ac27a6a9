#!/bin/bash
export TMPDIR=/tmp
rm -rf gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 1 --warmup 0 --cpu-steps 0 --no-profile-pass > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/tl/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_time" in r["Kernel_Name"] or "k_freq" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[2000]["Start_Timestamp"])
print("cols:", list(rows[0].keys())[:14])
for r in rows[2000:2024]:
    name = "T" if "k_time" in r["Kernel_Name"] else "F"
    print(name, r.get("Queue_Id"), r.get("Stream_Id"), "start %7.2f end %7.2f dur %5.2f us" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
rm -rf gpurun_out/tl
