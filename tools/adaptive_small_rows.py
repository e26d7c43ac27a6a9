import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from opticomlib_amd import _lib, devices, workloads
dt = 1.0 / (16 * 32e9)
for k, rows in ((12, 1), (12, 2), (11, 2)):
    n = 1 << k
    a = workloads.qpsk_field(n, seed=1, n_pol=2, power_w=5e-3)[:rows]
    D = devices.linear_operator(n, dt, 0.2, -21.7, 0.13)
    p = _lib.Plan(n, rows, _lib.C64); p.set_linear_operator(D)
    for rep in range(3):
        p.set_field(a); p.synchronize()
        t = time.perf_counter(); s, z, _ = p.propagate_adaptive(1.3, 40.0, 0.002, False); el = time.perf_counter() - t
    print(k, rows, "steps", s, "us/step %.2f" % (el / s * 1e6), "launches", p.last_propagate_ms()[1])
