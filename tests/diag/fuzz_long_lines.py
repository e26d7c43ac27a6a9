"""Random runs of lengths that are not powers of two above 65536 samples (round 6: the chirp-z line held as complex64 between float64 passes, on two lanes, a table of n
entries): fixed schedules (one or two step sizes: four launches per step), adaptive runs (five per step), one and two polarisations, lines of 2^18 ... 2^21 points.
Judge: the complex128 line of the same call (SSFM_CHIRP_HALF=0), which is 1e-13 from the float64 solution of its schedule -- the difference must stay below 2e-6; every
fourth case also against the oracle under the suite's bound.        python tests/diag/fuzz_long_lines.py [count] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import opticomlib_amd as oa
from opticomlib_amd import workloads, accuracy
from opticomlib_amd.typing import gv, optical_signal
from oracle import ssfm_numpy as orc
gv(**workloads.BENCH_GV)
count = int(sys.argv[1]) if len(sys.argv) > 1 else 48
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
rel = lambda u, v: float(np.abs(u - v).max() / np.abs(v).max())
worst, bad, t0 = 0.0, 0, time.time()
for i in range(count):
    lg = int(rng.integers(17, 21))                                   # the line: 2^(lg + 1) points
    n = int(rng.integers((1 << (lg - 1)) + 1, 1 << lg))
    if n & (n - 1) == 0: n += 1
    if i % 7 == 0: n = ((1 << int(rng.integers(13, 17))) - 1) * 16   # a PRBS word at 16 samples per bit
    n = max(n, 65537)
    npol = int(rng.integers(1, 3))
    a = workloads.qpsk_field(n - n % 16 + 16, seed=int(rng.integers(1 << 30)), n_pol=npol, power_w=float(rng.uniform(0.5e-3, 4e-3)))[:, :n]
    a = a if npol == 2 else a[0]
    fib = dict(alpha=float(rng.choice([0.0, 0.2, 0.35])), beta_2=float(rng.uniform(-25, 25)), beta_3=float(rng.choice([0.0, 0.13])), gamma=float(rng.uniform(0.5, 2.5)))
    mode = rng.choice(["fixed", "fixed", "adaptive"])
    if mode == "fixed":
        steps = int(rng.integers(3, 40))
        h = float(rng.uniform(0.1, 0.6))
        kw = dict(length=steps * h - float(rng.choice([0.0, 0.3 * h])), h=h, **fib)
    else:
        kw = dict(length=float(rng.uniform(2, 8)), phi_max=float(rng.uniform(0.004, 0.03)), **fib)
    os.environ.pop("SSFM_CHIRP_HALF", None)
    y = oa.FIBER(optical_signal(a), **kw).signal
    os.environ["SSFM_CHIRP_HALF"] = "0"
    w = oa.FIBER(optical_signal(a), **kw).signal
    os.environ.pop("SSFM_CHIRP_HALF", None)
    d = rel(y, w)
    line = f"{i:3d} n={n:7d} x {npol} {mode:8s} {str({k: round(v, 4) for k, v in kw.items()}):120s} vs complex128 line {d:.2e}"
    ok = d <= 2e-6 and not np.array_equal(y, w)
    if i % 4 == 0 and n * (40 if mode == "adaptive" else steps) <= 8e6:
        ref = orc.fiber_c64(a, gv.dt, **kw)
        nst = steps if mode == "fixed" else 100
        e = rel(y, ref)
        line += f"  vs oracle {e:.2e} (bound {accuracy.tol(nst):.1e})"
        ok = ok and e <= accuracy.tol(nst)
    worst = max(worst, d)
    bad += 0 if ok else 1
    print(line + ("" if ok else "   <-- VIOLATION"), flush=True)
print(f"{count} cases in {time.time() - t0:.0f} s: worst distance from the complex128 line {worst:.2e}, violations: {bad}")
