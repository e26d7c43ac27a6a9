"""A long adaptive run at 2^16 x 2 (complex64) for a rocprofv3 kernel trace."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from opticomlib_amd import _lib, devices, workloads
n = 1 << 16; dt = 1.0 / (16 * 32e9)
a = workloads.qpsk_field(n, seed=1, power_w=5e-3)
p = _lib.Plan(n, 2, _lib.C64); p.set_linear_operator(devices.linear_operator(n, dt, 0.2, -21.7, 0.13))
for rep in range(3):
    p.set_field(a); p.synchronize()
    t = time.perf_counter(); s, z, _ = p.propagate_adaptive(1.3, 40.0, 0.002, False); el = time.perf_counter() - t
print(f"{s} steps, {el/s*1e6:.2f} us/step, {p.last_propagate_ms()[1]} launches")
hs, _ = devices.step_schedule(200 * 0.1, 0.1)
p.set_field(a); p.propagate_fixed(1.3, hs); p.synchronize()
