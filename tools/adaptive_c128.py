"""Adaptive-step run of a complex128 plan at 2^20 x 2 (k_freq<double, FLY>: SSFM_EF=8 against 16)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from opticomlib_amd import _lib, devices, workloads
for k in (16, 20):
    n = 1 << k; dt = 1.0 / (16 * 32e9)
    a = workloads.qpsk_field(n, seed=1, power_w=10e-3)
    D = devices.linear_operator(n, dt, 0.2, -21.7, 0.13, _lib.C128)
    p = _lib.Plan(n, 2, _lib.C128); p.set_linear_operator(D)
    for rep in range(3):
        p.set_field(a); p.synchronize()
        t = time.perf_counter(); s, z, _ = p.propagate_adaptive(1.3, 80.0, 0.01, False); el = time.perf_counter() - t
    print(f"c128 2^{k} x 2: {s} adaptive steps in {el*1e3:.1f} ms -> {el/s*1e6:.1f} us/step (EF={os.environ.get('SSFM_EF', 'default')})")
    p.close()
