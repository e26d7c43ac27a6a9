"""For the worst cases of fuzz_many: is the distance between the HIP result and the complex64 oracle the HIP path's error or
the reference's own float32 rounding?  Re-runs a case and compares both with a float64 run of the same float32 schedule."""
import os, sys, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import opticomlib_amd as oa
from opticomlib_amd import workloads
from opticomlib_amd.typing import gv, optical_signal
from oracle import ssfm_numpy as orc

count = int(sys.argv[1]) if len(sys.argv) > 1 else 400
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
rng = np.random.default_rng(seed)
gv(**workloads.BENCH_GV)
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
    if n not in (15060, 13232):
        continue
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        zr, Ar = orc.fiber_c64(a, gv.dt, return_steps=True, **kw)
        y = oa.FIBER(optical_signal(a), **kw).signal
        # float64 arithmetic on the oracle's own float32 step schedule (fixed steps h_k = diff(zr))
        A = np.asarray(a, np.complex128)
        D = orc.linear_operator_c64(a.shape[-1], gv.dt, kw["alpha"], kw["beta_2"], kw["beta_3"]).astype(np.complex128) if hasattr(orc, "linear_operator_c64") else None
        if D is None:
            print("no linear_operator_c64 in the oracle"); break
        g = float(np.float32(kw["gamma"]))
        for hk in np.diff(zr.astype(np.float32)).astype(np.float64):
            Nh = 1j * g * np.abs(A) ** 2
            A = A * np.exp(hk / 2 * Nh); A = np.fft.fft(A); A = A * np.exp(D * hk); A = np.fft.ifft(A); A = A * np.exp(hk / 2 * Nh)
    ref = Ar[-1]
    pk = np.max(np.abs(ref))
    print(f"n {n} x {npol}, {len(zr) - 1} steps: HIP vs complex64 oracle {np.max(np.abs(y - ref)) / pk:.2e};  HIP vs float64 on the same schedule "
          f"{np.max(np.abs(y - A)) / pk:.2e};  complex64 oracle vs float64 {np.max(np.abs(ref - A)) / pk:.2e}")
