"""Where does the error of a complex64 chirp-z (Bluestein) split-step line come from?  (CPU only, NumPy emulation.)

One step of the one-launch line (csrc: k_medium_chirp) is, on a line of M = 2^k >= 2n - 1 points,
    FFT_M . H0 . IFFT_M  ->  x exp(D~ h)/n, zero from n up  ->  FFT_M . H1 . IFFT_M  ->  zero from n up, the nonlinear rotation of two half steps
(the chirps on either side of a step cancel against the neighbours': only the run's two ends carry them).  Every piece can be computed either the way a
complex64 kernel does it or exactly (float64, rounded once to complex64 where the kernel keeps complex64 data):
    fft   "c64":  numpy.fft in single precision (pocketfft: correctly rounded float32 twiddles, like the kernel's)
          "exact": the transform in complex128, its result rounded to complex64 -- a transform whose only error is the rounding of its output
    tab   "c64":  H0 / H1 rounded to complex64 (from double)          "exact": the product in complex128, rounded
    op    "c64":  exp(D~ h)/n rounded to complex64                     "exact": the product in complex128, rounded
Against the float64 solution of the same float32 schedule (exact length-n transforms).  The oracle's own distance (complex64 pocketfft on the length n itself)
is printed beside it.          python tests/diag/chirp_c64_sources.py  ->  profiles/r05_chirp_c64_sources.txt
"""
import os, sys, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from opticomlib_amd import workloads
from oracle import ssfm_numpy as orc

F32 = np.float32
C64, C128 = np.complex64, np.complex128


def tables(n):
    M = 1 << (2 * n - 2).bit_length()
    m = np.arange(n)
    ph = (m * m) % (2 * n)
    c = np.exp(-1j * np.pi * ph / n)                                 # forward chirp c_m = exp(-i pi m^2 / n)
    v = np.zeros(M, C128); v[:n] = np.conj(c); v[M - n + 1:] = np.conj(c[1:][::-1])
    H0 = np.fft.fft(v)                                               # transfer function of the convolution with conj(c)
    return M, c, H0, np.conj(H0)


def fft_(x, inverse, how):
    f = np.fft.ifft if inverse else np.fft.fft
    if how == "c64":
        return f(x.astype(C64))
    return f(x.astype(C128)).astype(C64)


def mul_(x, t, how):
    if how == "c64":
        return (x.astype(C64) * t.astype(C64)).astype(C64)
    return (x.astype(C128) * t).astype(C64)


def run_line(a0, dt, hs, fib, fft="c64", tab="c64", op="c64"):
    """The complex64 chirp line over the float32 schedule hs; returns the complex64 field."""
    n = a0.shape[-1]
    M, c, H0, H1 = tables(n)
    D = orc.linear_operator_c64(n, dt, fib["alpha"], fib["beta_2"], fib["beta_3"]).astype(C128)
    g = F32(fib["gamma"])
    A = a0.astype(C64)
    line = np.zeros(a0.shape[:-1] + (M,), C64)
    # head of the run: |A|^2, first half rotation, times the chirp
    P = (A.real * A.real + A.imag * A.imag).astype(F32)
    for k, h in enumerate(hs):
        h = F32(h)
        hh = F32(h * F32(0.5))
        if k == 0:
            rot = np.exp(1j * (hh * (g * P)).astype(np.float64)).astype(C64)
            x = (A * rot).astype(C64)
            line[..., :n] = mul_(x, c, "exact")                      # (the run's two ends: once per run)
        lin = np.exp(D * np.float64(h)) / n
        s = fft_(line, False, fft)
        s = mul_(s, H0, tab)
        y = fft_(s, True, fft)
        y[..., n:] = 0
        y[..., :n] = mul_(y[..., :n], lin, op)                       # y_k = X_k conj(c_k); the inverse wants X_k conj(c_k) too: the chirps cancel
        s = fft_(y, False, fft)
        s = mul_(s, H1, tab)
        y = fft_(s, True, fft)
        y[..., n:] = 0
        # y[:n] = A_time * c  (still chirped); the rotation commutes with the chirp
        Pn = (y[..., :n].real ** 2 + y[..., :n].imag ** 2).astype(F32)
        last = k == len(hs) - 1
        hn = F32(0) if last else F32(F32(hs[k + 1]) * F32(0.5))
        phi = (hh * (g * P)).astype(F32) + (F32(0) if last else (hn * (g * Pn)).astype(F32))
        y[..., :n] = (y[..., :n] * np.exp(1j * phi.astype(np.float64)).astype(C64)).astype(C64)
        P = Pn
        line = y
    return mul_(line[..., :n], np.conj(c), "exact")


def run_f64(a0, dt, hs, fib):
    n = a0.shape[-1]
    D = orc.linear_operator_c64(n, dt, fib["alpha"], fib["beta_2"], fib["beta_3"]).astype(C128)
    g = np.float64(F32(fib["gamma"]))
    A = a0.astype(C64).astype(C128)
    for h in hs:
        h = np.float64(F32(h)); hh = np.float64(F32(F32(h) * F32(0.5)))
        P = np.abs(A) ** 2
        A = A * np.exp(1j * g * P * hh)
        A = np.fft.ifft(np.fft.fft(A) * np.exp(D * h))
        A = A * np.exp(1j * g * P * hh)
    return A


def relmax(a, b):
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


CASES = [
    # the three cases of profiles/r04_final_fuzz.txt beyond 2e-5, then the family of tests/diag/chirp_medium_check.py
    ("fuzz 13232x1 gain", 13232, 1, 1e-2, dict(alpha=-0.24271232470155357, beta_2=-10.583432784716948, beta_3=0.0, gamma=-3.2622050235402775), 34.60768683611645, 1.0),
    ("fuzz 10426x1", 10426, 1, 1e-2, dict(alpha=0.015428355364570934, beta_2=-10.576241433628809, beta_3=0.3537441346881304, gamma=2.8669654493109538), 16.356927853605633, 0.25),
    ("smf 8176x2", 8176, 2, 4e-3, workloads.SMF, 50.0, 0.5),
    ("smf 32752x1", 32752, 1, 4e-3, workloads.SMF, 50.0, 0.5),
]

if __name__ == "__main__":
    dt = 1.0 / (workloads.BENCH_GV["sps"] * workloads.BENCH_GV["R"])
    out = ["# max|A - A_float64| / max|A_float64| at the end of the run; columns: the oracle (complex64 pocketfft on n itself) | the line all complex64 | exact transforms | exact H tables | exact operator | all three exact (only the data's roundings left)",
           "# case | steps | oracle | line c64 | fft exact | tab exact | op exact | all exact | line-vs-oracle (c64) | line-vs-oracle (all exact)"]
    for name, n, npol, pw, fib, length, h in CASES:
        a = workloads.qpsk_field(1 << max(6, (n - 1).bit_length()), seed=n % 997, n_pol=npol, power_w=pw)[:, :n]
        # the float32 schedule, as the reference walks it
        hs, z, h_ = [], F32(0), F32(min(F32(h), F32(length)))
        while z < F32(length):
            z = F32(z + h_); hs.append(h_); h_ = F32(min(h_, F32(length) - z))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            ref = run_f64(a, dt, hs, fib)
            orac = orc.fiber_c64(a, dt, length=length, h=h, **fib)
            res = {k: run_line(a, dt, hs, fib, **kw) for k, kw in (("c64", {}), ("fft", dict(fft="exact")), ("tab", dict(tab="exact")), ("op", dict(op="exact")),
                                                                   ("all", dict(fft="exact", tab="exact", op="exact")))}
        line = (f"{name:20s} | {len(hs):4d} | {relmax(orac, ref):.2e} | " + " | ".join(f"{relmax(res[k], ref):.2e}" for k in ("c64", "fft", "tab", "op", "all"))
                + f" | {relmax(res['c64'], orac):.2e} | {relmax(res['all'], orac):.2e}")
        print(line, flush=True)
        out.append(line)
    open(os.path.join(ROOT, "profiles", "r05_chirp_c64_sources.txt"), "w").write("\n".join(out) + "\n")
