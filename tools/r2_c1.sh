#!/bin/bash
# complex128 (C1): kernel split and SQ counters
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/c1ks; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c1ks -- python3 tools/c1_prof.py > /dev/null 2>&1
find gpurun_out/c1ks -name "*kernel_stats.csv" | head -1 | xargs -r head -8 | cut -c1-260 | tee gpurun_out/r2_c1_kernel_stats.csv
find gpurun_out/c1ks -name "*kernel_trace.csv" -delete
bash tools/gpu_pmc_c1.sh pmc_c1 | tee gpurun_out/r2_c1_sq.txt
SSFM_LANES=1 python tools/c1_prof.py; python - <<'PY'
import sys, time; sys.path.insert(0, '.')
import os
import numpy as np
from opticomlib_amd import _lib, devices, workloads
n = 1 << 20; dt = 1.0 / (16 * 32e9)
a = workloads.qpsk_field(n, seed=2024)
p = _lib.Plan(n, 2, _lib.C128); p.set_linear_operator(devices.linear_operator(n, dt, 0.2, -21.7, 0.13, _lib.C128)); p.set_field(a)
hs, _ = devices.step_schedule(1000, 1.0, _lib.C128)
for _ in range(2): p.propagate_fixed(1.3, hs)
p.synchronize()
print("c128 1000 steps: %.2f us/step" % (p.last_propagate_ms()[0] * 1e3 / 1000))
PY
