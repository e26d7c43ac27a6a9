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
import chirp_c64_sources as S


def cases(count, seed):
    """fuzz_many.py's random stream, case by case (the same draws in the same order)."""
    rng = np.random.default_rng(seed)
    for i in range(count):
        pow2 = rng.integers(0, 3) > 0
        n = 1 << int(rng.integers(8, 15)) if pow2 else int(rng.integers(2, 20000))
        npol = int(rng.integers(1, 3))
        sign = -1.0 if rng.integers(0, 4) == 0 else 1.0
        fib = dict(alpha=sign * float(rng.uniform(0, 0.5)), beta_2=sign * float(rng.uniform(-30, 30)),
                   beta_3=sign * float(rng.choice([0.0, rng.uniform(-0.5, 0.5)])), gamma=sign * float(rng.choice([0.0, rng.uniform(0.3, 4)])))
        length = float(rng.uniform(0.5, 40))
        kw = dict(length=length, **fib)
        if rng.integers(0, 2):
            kw["phi_max"] = float(rng.choice([0.005, 0.01, 0.05]))
        else:
            kw["h"] = float(rng.choice([length / 5.7, 0.25, 1.0, 2.0, length * 3]))
        amp = float(rng.choice([0.01, 0.03, 0.1]))
        if "phi_max" in kw or rng.integers(0, 2):
            m = 1 << max(6, (n - 1).bit_length())
            a = workloads.qpsk_field(m, seed=int(rng.integers(0, 1 << 30)), n_pol=npol, power_w=amp ** 2)[:, :n]
        else:
            a = (rng.standard_normal((npol, n)) + 1j * rng.standard_normal((npol, n))) * amp
        a = a[0] if npol == 1 else a
        yield i, n, npol, kw, a, pow2


if __name__ == "__main__":
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
    lo = int(os.environ.get("FUZZ_MIN_N", "2049"))
    gv(**workloads.BENCH_GV)
    print("# i n x pol steps mode | B = |gamma| sum_k h_k max|A_k|^2 [rad] | HIP-oracle HIP-f64 oracle-f64 | engine")
    for i, n, npol, kw, a, pow2 in cases(count, seed):
        if pow2 or n < lo:
            continue
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            zr, Ar = orc.fiber_c64(a, gv.dt, return_steps=True, **kw)
            y = oa.FIBER(optical_signal(a), **kw)
            hs = np.diff(zr.astype(np.float32)).astype(np.float32)
            fib = {k: kw.get(k, 0.0) for k in ("alpha", "beta_2", "beta_3", "gamma")}
            t = S.run_f64(np.atleast_2d(a), gv.dt, hs, fib).reshape(a.shape)
        pk = np.max(np.abs(t))
        B = abs(kw.get("gamma", 0.0)) * float(np.sum(hs * np.array([np.max(np.abs(Ar[k]) ** 2) for k in range(len(hs))])))
        print(f"{i:4d} {n:6d} x {npol} {len(hs):5d} {'adapt' if 'phi_max' in kw else 'fixed'} | B {B:8.3f} | {np.max(np.abs(y.signal - Ar[-1])) / pk:.2e} {np.max(np.abs(y.signal - t)) / pk:.2e} "
              f"{np.max(np.abs(Ar[-1] - t)) / pk:.2e} | {getattr(y, 'engine', '-')}", flush=True)
