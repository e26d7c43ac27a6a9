"""Per-step time of the chirp-z path at the lengths the reference's own generators produce (PRBS order 7 ... 11 at 16 samples per bit), fixed and adaptive (dev aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import opticomlib_amd as oa
from opticomlib_amd import workloads
from opticomlib_amd.typing import gv, optical_signal
gv(**workloads.BENCH_GV)
for n in (2032, 2048, 8176, 8192, 32752, 32768):
    a = workloads.qpsk_field(1 << 15, seed=1, power_w=5e-3)[:, :n]
    x = optical_signal(a)
    for kw, name in ((dict(length=50, h=0.5), "fixed, 100 steps"), (dict(length=40, phi_max=0.002), "adaptive")):
        kw.update(workloads.SMF)
        oa.FIBER(x, **kw)
        t = time.perf_counter(); z, A = oa.FIBER(x, return_steps=False, **kw), None; el = time.perf_counter() - t
        steps = 100 if "h" in kw else None
        if steps is None:
            zz, _ = oa.FIBER(x, return_steps=True, **kw); steps = len(zz) - 1
        print(f"n = {n} x 2, {name}: {el * 1e3:.2f} ms, {steps} steps -> {el / steps * 1e6:.1f} us per step", flush=True)
