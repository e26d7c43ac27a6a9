"""Per-step time against field size, fixed and adaptive step, single polarisation and dual (dev aid)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from opticomlib_amd import _lib, devices, workloads
dt = 1.0 / (16 * 32e9)
PREC = _lib.C128 if "c128" in sys.argv[1:] else _lib.C64          # python tools/small_n.py [c128]
for k in (10, 12, 14, 15, 16, 17, 18, 19, 20):
    n = 1 << k
    for pol in (1, 2):
        a = workloads.qpsk_field(n, seed=1, n_pol=2, power_w=5e-3)[:pol]
        D = devices.linear_operator(n, dt, 0.2, -21.7, 0.13, PREC)
        p = _lib.Plan(n, pol, PREC); p.set_linear_operator(D); p.set_field(a)
        hs, _ = devices.step_schedule(200 * 0.1, 0.1, PREC)
        p.propagate_fixed(1.3, hs); p.synchronize()
        t = time.perf_counter(); p.propagate_fixed(1.3, hs); p.synchronize(); tf = (time.perf_counter() - t) / hs.size
        p.set_field(a); p.synchronize()
        t = time.perf_counter(); s, z, _ = p.propagate_adaptive(1.3, 5.0, 0.002, False); ta = (time.perf_counter() - t) / max(s, 1)
        print(f"2^{k} x {pol}: fixed {tf*1e6:6.1f} us/step   adaptive {ta*1e6:6.1f} us/step ({s} steps)", flush=True)
        p.close()
