#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for v in base notwn notab nop notables nofft memonly; do
  echo -n "$v: "; SSFM_LIB=$ROOT/build/abl/_ssfm_$v.so python - <<'PY'
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from opticomlib_amd import _lib, devices, workloads
n = 1 << 20; dt = 1.0 / (16 * 32e9)
hs, _ = devices.step_schedule(125.0, 0.125)
D = devices.linear_operator(n, dt, 0.2, -21.7, 0.13)
fields = 4
os.environ["SSFM_LANES"] = "2"
a = np.concatenate([workloads.qpsk_field(n, seed=s) for s in range(fields)]).astype(np.complex64)
p = _lib.Plan(n, 2 * fields, _lib.C64); p.set_linear_operator(D); p.set_field(a)
p.propagate_fixed(1.3, hs); p.synchronize()
t = time.perf_counter()
for _ in range(2): p.propagate_fixed(1.3, hs)
p.synchronize()
el = (time.perf_counter() - t) / 2
print(f"{el/1000/fields*1e6:.2f} us per field-step")
PY
done
