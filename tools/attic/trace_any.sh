export TMPDIR=/tmp
cat > /tmp/any1.py <<'PY'
import sys, time; sys.path.insert(0, '.')
import numpy as np
import opticomlib_amd as oa
from opticomlib_amd import workloads
from opticomlib_amd.typing import gv, optical_signal
gv(**workloads.BENCH_GV)
n = 1000000
rng = np.random.default_rng(1)
a = (rng.standard_normal((2, n)) + 1j * rng.standard_normal((2, n))) * 0.03
oa.FIBER(optical_signal(a), length=10, h=0.5, **workloads.SMF).signal
PY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/anyt -- python3 /tmp/any1.py > /dev/null 2>&1
f=$(find gpurun_out/anyt -name "*kernel_stats.csv" | head -1)
python3 - $f <<PY
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r["Calls"]) >= 20: print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
find gpurun_out/anyt -name "*kernel_trace.csv" -delete
