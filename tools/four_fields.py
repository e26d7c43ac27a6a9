"""Per field-step time with FIELDS (default 4) C2-shaped fields resident in one plan (C3 at 8 / FIELDS GPUs)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from opticomlib_amd import _lib, devices, workloads

n = 1 << 20
dt = 1.0 / (16 * 32e9)
hs, _ = devices.step_schedule(125.0, 0.125)
D = devices.linear_operator(n, dt, 0.2, -21.7, 0.13)
fields = int(os.environ.get("FIELDS", "4"))
a = np.concatenate([workloads.qpsk_field(n, seed=s) for s in range(fields)]).astype(np.complex64)
p = _lib.Plan(n, 2 * fields, _lib.C64)
p.set_linear_operator(D)
p.set_field(a)
p.propagate_fixed(1.3, hs)
p.synchronize()
t = time.perf_counter()
for _ in range(2):
    p.propagate_fixed(1.3, hs)
p.synchronize()
el = (time.perf_counter() - t) / 2
print(f"{el / 1000 / fields * 1e6:.2f} us per field-step ({fields} fields)")
