"""Per-step time of the chirp-z path (lengths that are not powers of two) (dev aid)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import opticomlib_amd as oa
from opticomlib_amd import workloads
from opticomlib_amd.typing import gv, optical_signal
gv(**workloads.BENCH_GV)
for n in (3000, 65537, 1000000, (1 << 20) + 1, 1 << 21):
    rng = np.random.default_rng(1)
    a = (rng.standard_normal((2, n)) + 1j * rng.standard_normal((2, n))) * 0.03
    x = optical_signal(a)
    kw = dict(length=50, h=0.5, **workloads.SMF)
    oa.FIBER(x, **kw)
    t = time.perf_counter(); y = oa.FIBER(x, **kw); el = time.perf_counter() - t
    print(f"n = {n} x 2: 100 steps in {el*1e3:.1f} ms -> {el/100*1e6:.0f} us/step ({'fused engine' if n & (n-1) == 0 and n <= 1 << 22 else 'chirp-z'})", flush=True)
