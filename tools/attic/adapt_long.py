import os, sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from opticomlib_amd import _lib, devices, workloads
n = 1 << 20; dt = 1.0 / (16 * 32e9)
a = workloads.qpsk_field(n, seed=1, power_w=10e-3).astype(np.complex64)
D = devices.linear_operator(n, dt, 0.2, -21.7, 0.13)
p = _lib.Plan(n, 2, _lib.C64); p.set_linear_operator(D)
best = 1e9
for rep in range(4):
    p.set_field(a); p.synchronize()
    t = time.perf_counter(); s, z, _ = p.propagate_adaptive(1.3, 80.0, 0.002, False); el = time.perf_counter() - t
    best = min(best, el / s)
print(f"{s} adaptive steps -> {best*1e6:.2f} us/step")
