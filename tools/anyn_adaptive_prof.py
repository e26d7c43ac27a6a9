"""An adaptive chirp-z run of a long length for a rocprofv3 kernel trace (dev aid).  N=..."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import opticomlib_amd as oa
from opticomlib_amd import workloads
from opticomlib_amd.typing import gv, optical_signal
gv(**workloads.BENCH_GV)
n = int(os.environ.get("N", (1 << 20) + 1))
rng = np.random.default_rng(1)
a = ((rng.standard_normal((2, n)) + 1j * rng.standard_normal((2, n))) * 0.03).astype(np.complex64)
x = optical_signal(a)
kw = dict(length=40.0, **workloads.SMF)
oa.FIBER(x, **kw)
t = time.perf_counter(); y = oa.FIBER(x, **kw); el = time.perf_counter() - t
print(f"n = {n} x 2 adaptive: {el*1e3:.1f} ms")
