"""Per-step time of chirp-z runs (lengths that are not powers of two) from the difference of a short and a long run: fixed step and adaptive,
the one-launch engines of n <= 2048 (default) or the launch-per-pass loop (SSFM_CHIRP_SMALL=0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import opticomlib_amd as oa
from opticomlib_amd import workloads
from opticomlib_amd.typing import gv, optical_signal
gv(**workloads.BENCH_GV)


def best(x, kw, reps=3):
    oa.FIBER(x, **kw)
    el = []
    for _ in range(reps):
        t = time.perf_counter(); oa.FIBER(x, **kw); el.append(time.perf_counter() - t)
    return min(el)


for n in (127, 508, 1016, 2032, 3000):
    a = workloads.qpsk_field(1 << 15, seed=1, power_w=5e-3)[:, :n]
    x = optical_signal(a)
    t = {s: best(x, dict(length=0.5 * s, h=0.5, **workloads.SMF)) for s in (100, 1100)}
    line = f"n = {n} x 2: fixed {t[100] * 1e3:.2f} ms / 100 steps, {(t[1100] - t[100]) / 1000 * 1e6:.2f} us per further step"
    ta, st = {}, {}
    for L in (20, 120):
        kw = dict(length=L, phi_max=0.002, **workloads.SMF)
        ta[L] = best(x, kw)
        os.environ["SSFM_CHIRP_LOOP"] = "python"
        z, _ = oa.FIBER(x, return_steps=True, **kw)
        del os.environ["SSFM_CHIRP_LOOP"]
        st[L] = len(z) - 1
    print(line + f"; adaptive {ta[20] * 1e3:.2f} ms / {st[20]} steps, {(ta[120] - ta[20]) / (st[120] - st[20]) * 1e6:.2f} us per further step", flush=True)
