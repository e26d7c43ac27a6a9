"""Calibration data for the accuracy margin of the chirp-z lines: every case of fuzz_many's stream whose length is not a power of two, with
the three distances (HIP - oracle, HIP - float64, oracle - float64; max|d|/peak), the step count and the nonlinear phase the run accumulates.
    python tests/diag/fuzz_chirp_margin.py [count] [seed]     (env FUZZ_ROUTE=c64|c128|default: force the complex64 one-launch line / the complex128 line)"""
import os, sys, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import opticomlib_amd as oa
from opticomlib_amd import workloads
from opticomlib_amd.typing import gv, optical_signal
from oracle import ssfm_numpy as orc


from fuzz_cases import cases, truth_f64


if __name__ == "__main__":
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
    lo = int(os.environ.get("FUZZ_MIN_N", "2049"))
    gv(**workloads.BENCH_GV)
    if os.environ.get("FUZZ_ROUTE") == "c64":                       # every run on the one-launch complex64 lines, whatever its length in steps
        oa.devices._c64_line_has_margin = lambda steps: True
        oa.devices._C64_LINE_NO_MARGIN = (1 << 30, 1 << 30)
    print("# i n x pol steps mode | B = |gamma| sum_k h_k max|A_k|^2 [rad] | HIP-oracle HIP-f64 oracle-f64 | engine")
    for i, n, npol, kw, a, pow2 in cases(count, seed):
        if pow2 or n < lo:
            continue
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            zr, Ar = orc.fiber_c64(a, gv.dt, return_steps=True, **kw)
            y = oa.FIBER(optical_signal(a), **kw)
            hs = np.diff(zr.astype(np.float32)).astype(np.float32)
            t = truth_f64(a, gv.dt, hs, kw)
        pk = np.max(np.abs(t))
        B = abs(kw.get("gamma", 0.0)) * float(np.sum(hs * np.array([np.max(np.abs(Ar[k]) ** 2) for k in range(len(hs))])))
        print(f"{i:4d} {n:6d} x {npol} {len(hs):5d} {'adapt' if 'phi_max' in kw else 'fixed'} | B {B:8.3f} | {np.max(np.abs(y.signal - Ar[-1])) / pk:.2e} {np.max(np.abs(y.signal - t)) / pk:.2e} "
              f"{np.max(np.abs(Ar[-1] - t)) / pk:.2e} | {getattr(y, 'engine', '-')}", flush=True)
