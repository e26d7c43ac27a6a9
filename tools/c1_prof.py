import sys, time; sys.path.insert(0, '.')
import numpy as np
from opticomlib_amd import _lib, devices, workloads
n = 1 << 20; dt = 1.0 / (16 * 32e9)
a = workloads.qpsk_field(n, seed=2024)
p = _lib.Plan(n, 2, _lib.C128); p.set_linear_operator(devices.linear_operator(n, dt, 0.2, -21.7, 0.13, _lib.C128)); p.set_field(a)
hs, _ = devices.step_schedule(100, 1.0, _lib.C128)
for _ in range(3): p.propagate_fixed(1.3, hs)
p.synchronize()
