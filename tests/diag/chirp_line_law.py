"""The one-launch complex64 chirp-z line's distance from the float64 solution against its step count, beyond the 100 steps profiles/r05_chirp_margin.txt
covers: is 7.5e-7 x steps^0.75 (opticomlib_amd.accuracy.C64_LINE_LAW, from which the routing window 32 ... 758 is derived) still an upper envelope at
200 ... 2000 steps?      python tests/diag/chirp_line_law.py"""
import os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_cases as fc
import opticomlib_amd as oa
from opticomlib_amd import accuracy, workloads
from opticomlib_amd.typing import gv, optical_signal
from oracle import ssfm_numpy as orc

gv(**workloads.BENCH_GV)
oa.devices._c64_line_has_margin = lambda steps: True            # every run on the one-launch complex64 line, whatever its length in steps
oa.devices._C64_LINE_NO_MARGIN = (1 << 30, 1 << 30)
print("# n x pol, steps | HIP-float64 (float32 coefficients), law 7.5e-7 s^0.75, ratio | HIP-oracle, tol(steps), ratio | oracle-float64 | engine")
worst = 0.0
for n, npol, power in ((3000, 2, 4e-3), (8176, 2, 4e-3), (15060, 1, 2e-3), (32752, 1, 4e-3)):
    a = workloads.qpsk_field(1 << 16, seed=n % 997, power_w=power, n_pol=2)[:npol, :n]
    a = a[0] if npol == 1 else a
    for steps in (25, 50, 100, 200, 400, 800, 1600):
        if n > 10000 and steps > 800:
            continue
        kw = dict(length=steps * 0.125, h=0.125, **workloads.SMF)
        hs, _ = oa.devices.step_schedule(kw["length"], kw["h"])
        out = oa.FIBER(optical_signal(a), **kw)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            ref = orc.fiber_c64(a, gv.dt, **kw)
        t = fc.truth_f64(a, gv.dt, hs, kw)
        pk = float(np.max(np.abs(t)))
        e_ht = float(np.max(np.abs(out.signal - t))) / pk
        e_ho = float(np.max(np.abs(out.signal - ref))) / pk
        e_ot = float(np.max(np.abs(ref - t))) / pk
        law = accuracy.c64_line_error(len(hs))
        worst = max(worst, e_ht / law)
        print(f"{n:6d} x {npol} {len(hs):5d} | {e_ht:.2e} {law:.2e} {e_ht / law:5.2f} | {e_ho:.2e} {accuracy.tol(len(hs)):.2e} {e_ho / accuracy.tol(len(hs)):5.2f} | {e_ot:.2e} | {out.engine}", flush=True)
print(f"worst HIP-float64 / law: {worst:.2f}")
