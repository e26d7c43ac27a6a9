"""Per-step time of the one-XCD engines: power-of-two plans of 2^13 ... 2^17 samples (fixed, 1000 steps; adaptive) and the chirp-z lines of 8176 / 32752
samples (dev aid; SSFM_LIB selects the library)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import opticomlib_amd as oa
from opticomlib_amd import _lib, devices, workloads
from opticomlib_amd.typing import optical_signal, gv
gv(**workloads.BENCH_GV)
dt = gv.dt
out = []
for k, pol in ((13, 2), (14, 1), (14, 2), (15, 2), (16, 1), (16, 2), (17, 1)):
    n = 1 << k
    a = workloads.qpsk_field(n, seed=1, n_pol=2, power_w=5e-3)[:pol]
    D = devices.linear_operator(n, dt, 0.2, -21.7, 0.13, _lib.C64)
    p = _lib.Plan(n, pol, _lib.C64); p.set_linear_operator(D); p.set_field(a)
    hs, _ = devices.step_schedule(1000 * 0.1, 0.1, _lib.C64)
    p.propagate_fixed(1.3, hs); p.synchronize()
    best = 1e9
    for r in range(3):
        t = time.perf_counter(); p.propagate_fixed(1.3, hs); p.synchronize(); best = min(best, (time.perf_counter() - t) / hs.size)
    eng = p.last_run_info()["engine"]
    p.set_field(a); p.synchronize()
    t = time.perf_counter(); s, z, _ = p.propagate_adaptive(1.3, 20.0, 0.002, False); ta = (time.perf_counter() - t) / max(s, 1)
    out.append(f"2^{k}x{pol} {eng} {best*1e6:.2f} / adaptive {p.last_run_info()['engine']} {ta*1e6:.2f} ({s})")
    p.close()
for n, pol in ((3000, 2), (8176, 2), (32752, 2)):
    a = workloads.qpsk_field(1 << 16, seed=2, n_pol=2, power_w=4e-3)[:pol, :n]
    x = optical_signal(a)
    kw = dict(length=500.0, h=0.5, **workloads.SMF)
    oa.FIBER(x, **kw)
    best = 1e9
    for r in range(3):
        t = time.perf_counter(); oa.FIBER(x, **kw); best = min(best, (time.perf_counter() - t) / 1000)
    out.append(f"chirp {n}x{pol} {best*1e6:.2f}")
print(" | ".join(out))
