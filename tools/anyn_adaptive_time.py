"""Adaptive (h=None) chirp-z runs of long lengths: us per step (dev aid).  N=a,b,c"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import opticomlib_amd as oa
from opticomlib_amd import workloads
from opticomlib_amd.typing import gv, optical_signal
gv(**workloads.BENCH_GV)
for n in [int(v) for v in os.environ.get("N", "300001,1000000,1048577").split(",")]:
    rng = np.random.default_rng(1)
    a = ((rng.standard_normal((2, n)) + 1j * rng.standard_normal((2, n))) * 0.03).astype(np.complex64)
    x = optical_signal(a)
    kw = dict(length=float(os.environ.get("LENGTH", "40")), **workloads.SMF)
    z, _ = oa.FIBER(x, return_steps=True, **kw)               # (the step count; the run itself repeats bit for bit)
    steps = len(z) - 1
    del _
    oa.FIBER(x, **kw)
    t0 = time.perf_counter(); y = oa.FIBER(x, **kw); el = time.perf_counter() - t0
    print(f"n = {n} x 2 adaptive: {steps} steps in {el*1e3:.1f} ms -> {el / steps * 1e6:.1f} us per step (the call's fixed costs included)", flush=True)
