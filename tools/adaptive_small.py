import sys, time; sys.path.insert(0, '.')
import numpy as np
from opticomlib_amd import _lib, devices, workloads
n = 1 << 14; dt = 1.0 / (16 * 32e9)
a = workloads.qpsk_field(n, seed=1, power_w=5e-3)[:1].astype(np.complex64)
p = _lib.Plan(n, 1, _lib.C64); p.set_linear_operator(devices.linear_operator(n, dt, 0.2, -21.7, 0.13))
for rep in range(3):
    p.set_field(a); p.synchronize()
    t = time.perf_counter(); s, z, _ = p.propagate_adaptive(1.3, 40.0, 0.002, False); el = time.perf_counter() - t
    print(f"{s} steps in {el*1e3:.2f} ms -> {el/s*1e6:.1f} us/step")
