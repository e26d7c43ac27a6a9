"""Per-step time of a long fixed-step chirp-z run (SSFM_CHIRP_DEBUG=1 prints the enqueue / total split of the launch-per-pass loop) (dev aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import opticomlib_amd as oa
from opticomlib_amd import workloads
from opticomlib_amd.typing import gv, optical_signal
gv(**workloads.BENCH_GV)
for n in (127, 508, 1016, 2032, 32752):
    a = workloads.qpsk_field(1 << 15, seed=1, power_w=5e-3)[:, :n]
    x = optical_signal(a)
    for steps in (100, 1100):
        kw = dict(length=0.5 * steps, h=0.5); kw.update(workloads.SMF)
        oa.FIBER(x, **kw)
        t = time.perf_counter(); oa.FIBER(x, **kw); el = time.perf_counter() - t
        if steps == 100: t100 = el
    print(f"n = {n} x 2: {t100 * 1e3:.2f} ms for 100 steps, {el * 1e3:.2f} ms for 1100 -> {(el - t100) / 1000 * 1e6:.2f} us per step", flush=True)
