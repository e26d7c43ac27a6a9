"""How far are the complex64 results from the float64 solution -- ours (HIP) and the reference's (oracle)?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from opticomlib_amd import FIBER, gv, optical_signal, workloads
from oracle import ssfm_numpy as orc
gv(**workloads.BENCH_GV)
def rel(a, b): return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
for k, steps, h in ((16, 100, 1.0), (16, 1000, 0.125), (20, 1000, 0.125)):
    a = workloads.qpsk_field(1 << k, seed=2024)
    kw = dict(length=steps * h, h=h, **workloads.SMF)
    y64 = FIBER(optical_signal(a), **kw).signal
    y128 = FIBER(optical_signal(a), precision="complex128", **kw).signal
    msg = f"2^{k} x 2, {steps} steps: HIP c64 vs float64: {rel(y64, y128):.2e}"
    if k <= 16:
        r64 = orc.fiber_c64(a, gv.dt, **kw)
        msg += f";  reference c64 (oracle) vs float64: {rel(r64, y128):.2e};  HIP c64 vs reference c64: {rel(y64, r64):.2e}"
    print(msg, flush=True)
