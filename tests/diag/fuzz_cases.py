"""The random FIBER / DBP configurations of the stress runs (fuzz_many.py, fuzz_chirp_margin.py) and of the seeded slice the GPU suite runs
(tests/test_gpu_parity.py::test_fuzz_slice_...), and the float64 solution they are judged against.  Test infrastructure."""
import os, sys, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from opticomlib_amd import accuracy, workloads
from oracle import ssfm_numpy as orc

F32 = np.float32


def cases(count, seed, pow2_only=False, minlog2=8, maxlog2=14):
    """(i, n, npol, kw, a, pow2) -- power-of-two and other lengths, fixed and adaptive steps, both polarisation counts, loss and gain."""
    rng = np.random.default_rng(seed)
    for i in range(count):
        pow2 = rng.integers(0, 3) > 0 or pow2_only
        n = 1 << int(rng.integers(minlog2, maxlog2 + 1)) if pow2 else int(rng.integers(2, 20000))
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
            # band-limited (the adaptive rule h = phi_max / max|A|^2 is numerically chaotic for white noise: the maximum of a
            # full-band field decorrelates over ~0.06 km, so a 1e-7 difference in h grows ~20x per step -- in the reference too)
            m = 1 << max(6, (n - 1).bit_length())
            a = workloads.qpsk_field(m, seed=int(rng.integers(0, 1 << 30)), n_pol=npol, power_w=amp ** 2)[:, :n]
        else:
            a = (rng.standard_normal((npol, n)) + 1j * rng.standard_normal((npol, n))) * amp
        a = a[0] if npol == 1 else a
        yield i, n, npol, kw, a, pow2


def truth_f64(a, dt, hs, kw):
    """The float64 solution of the reference's problem: the same float32 coefficients, the same float32 schedule `hs`, the linear operator's argument
    D~ h as the reference forms it (a complex64 product, devices.py:1179 -- with a phase of thousands of radians its rounding is part of the problem, not
    of the noise) -- and everything else (transforms, exponentials, |A|^2, the rotations) in float64 / complex128."""
    n = np.shape(a)[-1]
    D = orc.linear_operator_c64(n, dt, kw.get("alpha", 0.0), kw.get("beta_2", 0.0), kw.get("beta_3", 0.0))
    g = np.float64(F32(kw.get("gamma", 0.0)))
    A = np.asarray(a).astype(np.complex64).astype(np.complex128)
    for h in hs:
        h = F32(h)
        hh = np.float64(F32(h / 2))                                   # (h_/2: exact in float32)
        lin = np.exp((D * h).astype(np.complex64).astype(np.complex128))
        P = np.abs(A) ** 2
        A = A * np.exp(1j * g * P * hh)
        A = np.fft.ifft(np.fft.fft(A, axis=-1) * lin, axis=-1)
        A = A * np.exp(1j * g * P * hh)
    return A


def truth_adaptive_f64(a, dt, kw):
    """The float64 solution of the reference's ADAPTIVE problem (devices.py:1172-1196): float32 coefficients, the step rule h = phi_max / (|gamma| max|A|^2) clamped to what
    is left, evaluated in float32 as the reference does -- on the maximum of the FLOAT64 field -- and every transform, exponential and rotation in float64 / complex128.
    Returns (z, A).  (truth_f64 with the differences of a run's float32 z log is NOT this: z_k+1 - z_k reproduces h_k only to an ulp of z, 2e-6 of a step here, which at
    28 rad of dispersion per km and band edge is 2e-5 of the field after 40 steps.)"""
    n = np.shape(a)[-1]
    D = orc.linear_operator_c64(n, dt, kw.get("alpha", 0.0), kw.get("beta_2", 0.0), kw.get("beta_3", 0.0))
    g32, phi32, L32 = F32(kw.get("gamma", 0.0)), F32(kw.get("phi_max", 0.01)), F32(kw["length"])
    g = np.float64(g32)
    A = np.asarray(a).astype(np.complex64).astype(np.complex128)
    rule = lambda A_, z_: F32(min(F32(phi32 / (np.abs(g32) * F32((np.abs(A_) ** 2).max()))), F32(L32 - z_)))
    h, z, zs = F32(min(F32(phi32 / (np.abs(g32) * F32((np.abs(A) ** 2).max()))), L32)), F32(0), [0.0]
    while z < L32:
        z = F32(z + h)
        hh = np.float64(F32(h / 2))
        lin = np.exp((D * h).astype(np.complex64).astype(np.complex128))
        P = np.abs(A) ** 2
        A = A * np.exp(1j * g * P * hh)
        A = np.fft.ifft(np.fft.fft(A, axis=-1) * lin, axis=-1)
        A = A * np.exp(1j * g * P * hh)
        zs.append(float(z))
        h = rule(A, z)
    return np.array(zs), A


def tol_of(steps):
    """SURVEY.md 8(c) states 2e-5 at 100 steps and 3e-4 at 1000: ONE bound continuous in the step count (opticomlib_amd.accuracy.tol: flat up to 100
    steps, the log-log line between the two points, proportional to the steps beyond).  Rounds 3-5 used a step function (3e-4 from step 101)."""
    return accuracy.tol(steps)


def judge(y, oracle_out, truth, steps):
    """(ok, HIP-oracle, HIP-float64, oracle-float64), distances as max|d| / peak.  THE BUILDER'S RESTATEMENT of the contract (SURVEY.md 8(c) only states
    the plain bound against the reference's own run; callers report the count beyond the plain bound `e_ho <= tol` beside this one: `plain_ok`).
    The restated bound, round 5, has two halves:
      (1) against the float64 solution: within the tolerance, or at most 1.5 x as far from it as the oracle is;
      (2) against the oracle: within the tolerance, or -- where the oracle's own distance from the float64 solution leaves no room for that -- within
          2.5 x that distance (what (1) allows the two to be apart at most).
    The reference's complex64 run is up to 2.6e-5 from the float64 solution after < 100 steps for awkward lengths (large prime factors, where pocketfft
    itself goes through Bluestein) and strong nonlinearity: no computation, however exact, is then within 2e-5 of it (three such cases in 400,
    profiles/r05_final_fuzz.txt), and the HIP result must be about as close to the truth as the reference is."""
    pk = max(float(np.max(np.abs(truth))), 1e-30)
    e_ho = float(np.max(np.abs(y - oracle_out))) / pk
    e_ht = float(np.max(np.abs(y - truth))) / pk
    e_ot = float(np.max(np.abs(oracle_out - truth))) / pk
    tol = tol_of(steps)
    ok = e_ht <= max(tol, 1.5 * e_ot) and e_ho <= max(tol, 2.5 * e_ot)
    return ok, e_ho, e_ht, e_ot


def plain_ok(e_ho, steps):
    """The contract as SURVEY.md 8(c) states it: within tol(steps) of the reference's own complex64 run -- nothing else.  Reported by the stress runs as a
    separate count so that rounds stay comparable."""
    return e_ho <= tol_of(steps)


def run_case(oa, gv, optical_signal, kw, a):
    """One configuration on the HIP path, the oracle and the float64 solution: (engine, steps, judge(...))."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        zr, Ar = orc.fiber_c64(a, gv.dt, return_steps=True, **kw)
        out = oa.FIBER(optical_signal(a), **kw)
        hs = np.diff(zr.astype(np.float32)).astype(np.float32)
        t = truth_f64(a, gv.dt, hs, kw)
    return getattr(out, "engine", "-"), len(hs), judge(out.signal, Ar[-1], t, len(hs))
