"""Would the padded (chirp-z / Bluestein) transform of lengths that are not powers of two hold the complex64 tolerance if it ran in complex64 itself?
VERDICT r03 item 7(c): "complex64 padded transform for complex64 callers only if the N = 100003 / 12345 cases stay < 2e-5 at 100 steps; commit the error numbers
either way".  A NumPy emulation of the device path (no GPU needed): one split step = half nonlinear rotation, DFT_N by Bluestein on a power-of-two line of
M >= 2N - 1 points, times exp(D~ h), inverse DFT_N by Bluestein, half rotation; the padded transforms either in complex128 (what opticomlib_amd/csrc/chirpz.hip does)
or in complex64 (numpy.fft keeps single precision under NumPy 2), chirps and exp(D~ h) generated in double and rounded once -- the best a complex64 kernel could do.
Compared with the float64 solution of the same float32 step schedule (oracle.fiber_c128-like reference built from exact length-N transforms in complex128).

    python tests/diag/chirpz_c64_error.py            -> profiles/r04_chirpz_c64_error.txt
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from opticomlib_amd import workloads
from oracle import ssfm_numpy as orc


def bluestein(x, sign, dtype):
    """DFT (sign=-1) / unnormalised inverse DFT (sign=+1) of the last axis by the chirp-z identity on M = 2^k >= 2N - 1 points, padded transforms in `dtype`."""
    n = x.shape[-1]
    M = 1 << int(np.ceil(np.log2(2 * n - 1)))
    m = np.arange(n)
    ph = (m * m) % (2 * n)                                          # exact reduction of pi m^2 / n
    c = np.exp(sign * 1j * np.pi * ph / n)                          # chirp, double
    v = np.zeros(M, np.complex128); v[:n] = np.conj(c); v[M - n + 1:] = np.conj(c[1:][::-1])
    V = np.fft.fft(v).astype(dtype)                                 # transfer function of the chirp convolution, rounded once
    a = np.zeros(x.shape[:-1] + (M,), dtype)
    a[..., :n] = (x * c.astype(dtype)).astype(dtype)
    y = np.fft.ifft(np.fft.fft(a).astype(dtype) * V).astype(dtype)
    return (y[..., :n] * c.astype(dtype)).astype(dtype)


def run(n, steps, dtype, h=0.5):
    dt = 1.0 / (workloads.BENCH_GV["sps"] * workloads.BENCH_GV["R"])
    a0 = workloads.qpsk_field(1 << int(np.ceil(np.log2(n))), seed=n % 1000, power_w=4e-3)[:, :n]
    D = orc.linear_operator_c64(n, dt, 0.2, -21.7, 0.13).astype(np.complex128)       # the float32 coefficients, as every path uses them
    lin = np.exp(D * np.float32(h))
    g = np.float64(np.float32(1.3)); hh = np.float64(np.float32(h)) / 2
    # float64 solution with exact length-N transforms
    A = a0.astype(np.complex128)
    for _ in range(steps):
        P = np.abs(A) ** 2
        A = A * np.exp(1j * g * P * hh)
        A = np.fft.ifft(np.fft.fft(A) * lin)
        A = A * np.exp(1j * g * P * hh)
    ref = A
    A = a0.astype(dtype)
    rdt = np.float32 if dtype == np.complex64 else np.float64
    for _ in range(steps):
        P = (np.abs(A) ** 2).astype(rdt)
        rot = np.exp(1j * (g * P.astype(np.float64) * hh)).astype(dtype)
        A = (A * rot).astype(dtype)
        S = bluestein(A, -1, dtype)
        S = (S * lin.astype(dtype)).astype(dtype)
        A = (bluestein(S, +1, dtype) / rdt(n)).astype(dtype)
        A = (A * rot).astype(dtype)
    return float(np.max(np.abs(A - ref)) / np.max(np.abs(ref)))


if __name__ == "__main__":
    out = ["# max|A - A_float64| / max|A_float64| after `steps` split steps (h = 0.5 km, SMF, 4 mW QPSK-like field), padded chirp-z transforms in complex128 / complex64",
           "# tolerance of the complex64 path: 2e-5 at 100 steps (SURVEY.md 8(c)); the complex64 ORACLE itself sits 5e-6 ... 2e-5 from the float64 solution there",
           "#       N  steps   chirp-z in complex128   chirp-z in complex64"]
    for n in (3000, 12345, 100003):
        for steps in (10, 100):
            e128 = run(n, steps, np.complex128)
            e64 = run(n, steps, np.complex64)
            out.append(f"{n:9d} {steps:6d}   {e128:22.3e} {e64:22.3e}")
            print(out[-1], flush=True)
    open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "profiles", "r04_chirpz_c64_error.txt"), "w").write("\n".join(out) + "\n")
