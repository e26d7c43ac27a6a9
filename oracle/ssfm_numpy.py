"""NumPy restatement of the reference's split-step Fourier path.  TEST INFRASTRUCTURE.

This module restates, on plain arrays, what ``opticomlib.devices.FIBER`` / ``DBP`` /
``DM`` compute (reference ``opticomlib/devices.py``; line numbers are given on every
function).  It exists to *check* the HIP path and to be the timed CPU baseline of
``bench.py``; the product package never imports it.

Parity status: PINNED -- ``tests/test_oracle_golden.py`` compares every function here
bit-for-bit with outputs captured from an import of the reference under the same
NumPy (fixtures ``tests/golden/*.npz``, generator ``tests/golden/make_golden.py``).

Number types follow the reference exactly (semantics (ii) of SURVEY.md N1, NumPy >= 2):
all fibre coefficients, the frequency grid, the step size ``h`` and the position ``z``
are float32, the field is complex64, ``numpy.fft`` keeps single precision.  The
arithmetic *order* of every expression is the reference's, because float32 results
depend on it.  ``fiber_c128`` is the float64 twin (reference ``devices.py:2440-2486``).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def angular_frequency(n: int, dt: float) -> np.ndarray:
    """rad/s grid in FFT order -- reference ``typing.py:1641`` (``fftfreq(size, gv.dt)*2*pi``)."""
    return np.fft.fftfreq(n, dt) * 2 * np.pi


# --------------------------------------------------------------------------- complex64
class _Coeffs32:
    """float32 0-d coefficients -- reference ``devices.py:1137-1142``."""

    def __init__(self, length, alpha, beta_2, beta_3, gamma, phi_max):
        self.alpha = np.array(alpha / 4.343, dtype=F32)   # dB/km -> 1/km
        self.beta_2 = np.array(beta_2, dtype=F32)
        self.beta_3 = np.array(beta_3, dtype=F32)
        self.gamma = np.array(gamma, dtype=F32)
        self.length = np.array(length, dtype=F32)
        self.phi_max = np.array(phi_max, dtype=F32)


def linear_operator_c64(n, dt, alpha, beta_2, beta_3) -> np.ndarray:
    """D~(w) as complex64 in FFT order -- reference ``devices.py:1144-1145``.

    ``alpha`` in dB/km, ``beta_2`` ps^2/km, ``beta_3`` ps^3/km; result in 1/km.
    """
    c = _Coeffs32(0.0, alpha, beta_2, beta_3, 0.0, 0.0)
    w = np.asarray(angular_frequency(n, dt) * 1e-12, dtype=F32)      # rad/ps
    return -c.alpha / 2 + 1j / 2 * c.beta_2 * w**2 + 1j / 6 * c.beta_3 * w**3


def first_step_c64(A, c: _Coeffs32, h):
    """Initial step size -- reference ``devices.py:1155-1159``."""
    if h is None:
        if (c.beta_2 == 0 and c.beta_3 == 0) or c.gamma == 0:
            h_ = c.length
        else:
            h_ = c.phi_max / (np.abs(c.gamma) * (np.abs(A) ** 2)).max()
    else:
        h_ = np.array(h, dtype=F32)
    return np.array(min(h_, c.length), dtype=F32)


def ssfm_step_c64(A, D, gamma, h_):
    """One symmetric step NL/2 - L - NL/2 with the *stale* N^ -- reference ``devices.py:1175-1181``."""
    N_hat = 1j * gamma * np.abs(A) ** 2
    A = A * np.exp(h_ / 2 * N_hat)
    A = np.fft.fft(A)
    A = A * np.exp(D * h_)
    A = np.fft.ifft(A)
    A = A * np.exp(h_ / 2 * N_hat)
    return A


def fiber_c64(field, dt, length, alpha=0.0, beta_2=0.0, beta_3=0.0, gamma=0.0,
              phi_max=0.01, h=None, return_steps=False, max_steps=None):
    """``FIBER`` on a plain array -- reference ``devices.py:1113-1206``.

    ``field`` is ``signal + noise`` (reference ``typing.py:1596``), shape ``(N,)`` or
    ``(2, N)``, any dtype; ``dt`` is the sample period in seconds (the reference reads
    the global ``gv.dt``).  Returns the complex64 output field, or with
    ``return_steps`` the pair ``(z float64 (S+1,), A_z complex64 (S+1, [2,] N))``.
    ``max_steps`` (not in the reference) stops early; used only to time a bounded
    sample of a long run.
    """
    c = _Coeffs32(length, alpha, beta_2, beta_3, gamma, phi_max)
    n = np.shape(field)[-1]
    D = linear_operator_c64(n, dt, alpha, beta_2, beta_3)
    A = np.asarray(field, dtype=np.complex64)

    if return_steps:
        z_list = [0.0]
        A_list = [A.copy()]

    h_ = first_step_c64(A, c, h)
    z = np.array(0, dtype=F32)
    steps = 0
    while z < c.length:
        z += h_
        A = ssfm_step_c64(A, D, c.gamma, h_)
        steps += 1
        if return_steps:
            z_list.append(z.copy())
            A_list.append(A.copy())
        if h is None:                                       # devices.py:1193-1194
            h_ = c.phi_max / (np.abs(c.gamma) * (np.abs(A) ** 2)).max()
        h_ = np.array(min(h_, c.length - z), dtype=F32)     # devices.py:1196
        if max_steps is not None and steps >= max_steps:
            break

    if return_steps:
        return np.array(z_list), np.array(A_list)
    return A


def fiber_c64_tidy(field, dt, length, alpha=0.0, beta_2=0.0, beta_3=0.0, gamma=0.0, h=1.0, max_steps=None):
    """The same fixed-step arithmetic written the way a CPU user who cares about speed would (SURVEY.md 8(d),
    "fair many-core" baseline): ``exp(D~ h)`` tabulated per distinct step size instead of recomputed every
    step, the nonlinear factor ``exp(h/2 N^)`` evaluated once per step and applied twice, products in place.
    Every value is produced by the same expression as in :func:`ssfm_step_c64`, so the result is bit-identical
    to :func:`fiber_c64`; only the amount of work differs."""
    c = _Coeffs32(length, alpha, beta_2, beta_3, gamma, 0.01)
    n = np.shape(field)[-1]
    D = linear_operator_c64(n, dt, alpha, beta_2, beta_3)
    A = np.array(field, dtype=np.complex64)
    tables = {}
    h_ = first_step_c64(A, c, h)
    z = np.array(0, dtype=F32)
    steps = 0
    while z < c.length:
        z += h_
        key = h_.tobytes()
        if key not in tables:
            tables[key] = np.exp(D * h_)
        rot = np.exp(h_ / 2 * (1j * c.gamma * np.abs(A) ** 2))
        A *= rot
        A = np.fft.fft(A)
        A *= tables[key]
        A = np.fft.ifft(A)
        A *= rot
        steps += 1
        h_ = np.array(min(h_, c.length - z), dtype=F32)
        if max_steps is not None and steps >= max_steps:
            break
    return A


def dbp_c64(field, dt, length, alpha=0.0, beta_2=0.0, beta_3=0.0, gamma=0.0,
            phi_max=0.01, h=None, return_steps=False):
    """``DBP`` = ``FIBER`` with negated operators -- reference ``devices.py:1280-1283``."""
    return fiber_c64(field, dt, length, alpha=-alpha, beta_2=-beta_2, beta_3=-beta_3,
                     gamma=-gamma, phi_max=phi_max, h=h, return_steps=return_steps)


def step_schedule_c64(length, h) -> np.ndarray:
    """float32 step sizes of a fixed-``h`` run -- reference ``devices.py:1158-1161,1173,1196``.

    ``z`` accumulates in float32, so e.g. ``length=100, h=0.1`` yields 1001 steps.
    """
    L = np.array(length, dtype=F32)
    h_ = np.array(min(np.array(h, dtype=F32), L), dtype=F32)
    z = np.array(0, dtype=F32)
    out = []
    while z < L:
        z += h_
        out.append(F32(h_))
        h_ = np.array(min(h_, L - z), dtype=F32)
    return np.array(out, dtype=F32)


# --------------------------------------------------------------------------- DM (complex128)
def dm_transfer(n, dt, D):
    """``H(w) = exp(+1j w^2 D/2)``, D in ps^2, FFT order -- reference ``devices.py:1025-1027``."""
    D = D * 1e-12**2
    return np.exp(1j * angular_frequency(n, dt) ** 2 * D / 2)


def dm_c128(signal, dt, D, noise=None):
    """``DM`` -- reference ``devices.py:1019-1035``; signal and noise are filtered
    separately (``typing.py:1444-1449``, ``:1337-1344``).  Returns ``(signal, noise|None)``."""
    n = np.shape(signal)[-1]
    H = dm_transfer(n, dt, D)
    out_s = np.fft.ifft(np.fft.fft(np.asarray(signal), axis=-1) * H, axis=-1)
    out_n = None
    if noise is not None:
        out_n = np.fft.ifft(np.fft.fft(np.asarray(noise), axis=-1) * H, axis=-1)
    return out_s, out_n


# --------------------------------------------------------------------------- complex128 twin
def linear_operator_c128(n, dt, alpha, beta_2, beta_3) -> np.ndarray:
    """float64 D~ -- reference ``devices.py:2440-2442``."""
    alpha = alpha / 4.343
    w = angular_frequency(n, dt) * 1e-12
    return -alpha / 2 + 1j / 2 * beta_2 * w**2 + 1j / 6 * beta_3 * w**3


def fiber_c128(field, dt, length, alpha=0.0, beta_2=0.0, beta_3=0.0, gamma=0.0,
               phi_max=0.01, h=None, max_steps=None):
    """The SSFM loop in float64 / complex128.

    Operator order and the stale-N^ rule are the reference's float64 twin loop
    (``devices.py:2461-2470``); step control follows ``FIBER`` (``devices.py:1155-1159,
    1193-1196``: ``abs(gamma)``, maximum over both polarisations, ``h = min(h, L - z)``)
    with every float32 cast replaced by float64.  For fixed, exactly representable
    ``h`` both schedules coincide, which is what the golden check uses.
    """
    n = np.shape(field)[-1]
    D = linear_operator_c128(n, dt, alpha, beta_2, beta_3)
    A = np.asarray(field, dtype=np.complex128)
    length = float(length)
    if h is None:
        if (beta_2 == 0 and beta_3 == 0) or gamma == 0:
            h_ = length
        else:
            h_ = phi_max / (abs(gamma) * (np.abs(A) ** 2)).max()
    else:
        h_ = float(h)
    h_ = min(h_, length)
    z = 0.0
    steps = 0
    while z < length:
        z += h_
        N_hat = 1j * gamma * np.abs(A) ** 2
        A = A * np.exp(h_ / 2 * N_hat)
        A = np.fft.fft(A)
        A = A * np.exp(D * h_)
        A = np.fft.ifft(A)
        A = A * np.exp(h_ / 2 * N_hat)
        steps += 1
        if h is None:
            h_ = phi_max / (abs(gamma) * (np.abs(A) ** 2)).max()
        h_ = min(h_, length - z)
        if max_steps is not None and steps >= max_steps:
            break
    return A
