#!/bin/bash
# A/B two library builds, interleaved: single field (bench) and 4 fields resident.  usage: ab.sh libA libB
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for r in 1 2; do for L in "$@"; do
  echo -n "$(basename $L) single: "; SSFM_LIB=$L python $ROOT/bench.py --steps 4 --warmup 1 --cpu-steps 0 --no-profile-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us/step'%d['us_per_ssfm_step'])"
  echo -n "$(basename $L) 4 fields: "; SSFM_LIB=$L python - <<'PY'
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from opticomlib_amd import _lib, devices, workloads
n = 1 << 20; dt = 1.0 / (16 * 32e9)
hs, _ = devices.step_schedule(125.0, 0.125)
D = devices.linear_operator(n, dt, 0.2, -21.7, 0.13)
fields = 4
a = np.concatenate([workloads.qpsk_field(n, seed=s) for s in range(fields)]).astype(np.complex64)
p = _lib.Plan(n, 2 * fields, _lib.C64); p.set_linear_operator(D); p.set_field(a)
p.propagate_fixed(1.3, hs); p.synchronize()
t = time.perf_counter()
for _ in range(2): p.propagate_fixed(1.3, hs)
p.synchronize()
el = (time.perf_counter() - t) / 2
print(f"{el/1000/fields*1e6:.2f} us per field-step")
PY
done; done
