"""The fixed costs of an adaptive run (round 5): 2^20 x 2 complex64, runs of about 46, 131 and 652 steps; wall clock per run, best of five, and the launches
of the run.    [SSFM_LIB=...] python tools/attic/adaptive_fixed_costs.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from opticomlib_amd import _lib, devices, workloads
n = 1 << 20; dt = 1.0 / (16 * 32e9)
a = workloads.qpsk_field(n, seed=1, power_w=10e-3).astype(np.complex64)
D = devices.linear_operator(n, dt, 0.2, -21.7, 0.13)
p = _lib.Plan(n, 2, _lib.C64); p.set_linear_operator(D)
out = []
for length, phi in ((20.0, 0.01), (80.0, 0.01), (80.0, 0.002)):
    best, steps = 1e9, 0
    for rep in range(6):
        p.set_field(a); p.synchronize()
        t = time.perf_counter(); steps, z, _ = p.propagate_adaptive(1.3, length, phi, False); el = time.perf_counter() - t
        if rep: best = min(best, el)
    out.append(f"{steps} steps {best * 1e3:.3f} ms ({best / steps * 1e6:.1f} us per step, {p.last_propagate_ms()[1]} launches)")
print(" | ".join(out))
