"""Per-step cost of a long chirp-z run without the call's fixed costs: the slope between a 100-step and a 400-step run (dev aid).  N=... POL=..."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import opticomlib_amd as oa
from opticomlib_amd import workloads
from opticomlib_amd.typing import gv, optical_signal
gv(**workloads.BENCH_GV)
for n in [int(v) for v in os.environ.get("N", str((1 << 20) + 1)).split(",")]:
    pol = int(os.environ.get("POL", "2"))
    rng = np.random.default_rng(1)
    a = ((rng.standard_normal((pol, n)) + 1j * rng.standard_normal((pol, n))) * 0.03).astype(np.complex64)
    x = optical_signal(a if pol == 2 else a[0])
    t = {}
    for steps in (100, 400):
        kw = dict(length=steps * 0.5, h=0.5, **workloads.SMF)
        oa.FIBER(x, **dict(kw, length=1.0))
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter(); oa.FIBER(x, **kw); best = min(best, time.perf_counter() - t0)
        t[steps] = best
    print(f"n = {n} x {pol}: {(t[400] - t[100]) / 300 * 1e6:7.1f} us per step (slope), fixed {((t[100] * 4 - t[400]) / 3) * 1e3:6.2f} ms per call", flush=True)
