"""``FIBER`` / ``DBP`` / ``DM`` with the reference's call signatures, running on MI355X.

Host-side mirror of reference ``opticomlib/devices.py:1038-1206`` (FIBER), ``:1209-1283``
(DBP) and ``:945-1035`` (DM): same argument names, defaults, units, return types and the
same ``TypeError``.  All field arithmetic happens in the HIP library behind the C ABI
(``include/ssfm_amd.h``); this module only prepares the O(N) operator coefficients and the
float32 step schedule exactly the way the reference's prologue does, and wraps the result.

There is deliberately NO CPU fallback: sizes the HIP path does not cover raise
``ValueError``; a missing library or device raises ``SsfmError``.
"""
from __future__ import annotations

import os
import threading
import functools
import time
from collections import OrderedDict

import numpy as np

from . import _lib, accuracy
from .typing import NULL, binary_sequence, electrical_signal, gv, optical_signal

_F32 = np.float32
_CACHE_LOCK = threading.RLock()       # guards the module's small caches (plans, operators, grid powers, chirps) against concurrent callers
_PLANS: "OrderedDict[tuple, _lib.Plan]" = OrderedDict()
_MAX_PLANS = 12           # a 2^20 x 2 plan holds ~0.1 GB of the 288 GB; a link script cycles through a handful of shapes


def default_device() -> int:
    """Device index of this process: ``LOCAL_RANK`` under torchrun, else 0."""
    return int(os.environ.get("LOCAL_RANK", "0")) % max(_lib.device_count(), 1)


def get_plan(n: int, batch: int, precision: int, device=None) -> _lib.Plan:
    """Plans own all device buffers; keep the few most recent ones alive.

    Eviction only DROPS the cache's reference: the plan's buffers are freed when its last user lets go of it
    (``Plan.__del__``), never under a thread that was handed the plan a moment ago and has not taken its lock yet."""
    device = default_device() if device is None else int(device)
    key = (device, int(n), int(batch), int(precision))
    with _CACHE_LOCK:
        plan = _PLANS.pop(key, None)
        if plan is None or plan.closed:
            plan = _lib.Plan(n, batch, precision, device)
        _PLANS[key] = plan
        while len(_PLANS) > _MAX_PLANS:
            _PLANS.popitem(last=False)
    return plan


def release_plans():
    with _CACHE_LOCK:
        plans = list(_PLANS.values())
        _PLANS.clear()
    for p in plans:
        with p.lock:
            p.close()


def _tag(*key) -> int:
    """Non-zero 64-bit label of what a caller stages in a plan (``Plan.set_tag`` / ``Plan.tag``): the plan itself
    clears the label whenever the buffer is overwritten, so a matching label means the content is still there.

    A digest of the exact values, NOT Python's ``hash``: ``hash(-1.0) == hash(-2.0)`` in CPython, so a hash-labelled
    plan took ``FIBER(beta_2=-2)`` for the ``FIBER(beta_2=-1)`` it had staged.  Floats enter by their bit patterns
    (``-0.0`` and ``0.0``, ``1`` and ``1.0`` are different keys: a spurious miss only re-stages)."""
    import hashlib
    import struct
    h = hashlib.blake2b(digest_size=8)
    for k in key:
        if isinstance(k, float):
            h.update(b"f" + struct.pack("<d", k))
        elif isinstance(k, (int, np.integer)):
            h.update(b"i" + int(k).to_bytes(16, "little", signed=True))
        elif isinstance(k, str):
            h.update(b"s" + k.encode() + b"\0")
        else:
            raise TypeError(f"_tag: unsupported key component {type(k).__name__}")
    return int.from_bytes(h.digest(), "little") | 1


def _is_fast_size(n: int, precision: int) -> bool:
    """Power of two within the fused two-kernel engine's range."""
    lo, hi = _lib.supported_log2n(precision)
    return (1 << lo) <= n <= (1 << hi) and n & (n - 1) == 0


def _check_size(n: int, precision: int):
    """Any length from 2 up to 2^21 samples per polarisation runs on the GPU, and powers of two up to 2^24: powers of two in [2^8, 2^22] through
    the fused engine, 2^23 and 2^24 as split plans (round 6), every other length through the chirp-z path on a power-of-two plan of >= 2n - 1 <= 2^22 points."""
    if _is_fast_size(n, precision):
        return
    _, hi = _lib.supported_log2n(_lib.C128, direct=True)
    _, top = _lib.supported_log2n(_lib.C128)
    if n < 2 or 2 * n - 1 > (1 << hi):
        raise ValueError(
            f"the MI355X fibre path takes 2 ... 2^{hi - 1} samples per polarisation (powers of two up to 2^{top}), "
            f"got {n} (there is no CPU fallback)")


def _precision_code(precision) -> int:
    p = np.dtype(precision)
    if p == np.complex64:
        return _lib.C64
    if p == np.complex128:
        return _lib.C128
    raise ValueError("precision must be complex64 or complex128")


# ------------------------------------------------------------------ device residency
# Results of FIBER / DBP / DM / BPF / LPF stay in GPU memory (``signal`` / ``noise`` backed by a
# ``_lib.DeviceArray``) until somebody reads them on the host; the next device call of a chain takes them
# from there.  ``KEEP_ON_DEVICE = False`` restores eager host arrays.
KEEP_ON_DEVICE = True


def _on_device(a) -> bool:
    return isinstance(a, _lib.DeviceArray)


def _dev_array(x, dtype, dev: int) -> "_lib.DeviceArray":
    """``x`` (host array or DeviceArray) as a DeviceArray of ``dtype`` on GPU ``dev``; no transfer and no
    copy when it already is one."""
    dtype = np.dtype(dtype)
    if _on_device(x):
        if x.device != dev:                                 # other GPU: through the host
            return _lib.DeviceArray.from_host(x.to_host(), dtype, dev)
        return x if x.dtype == dtype else x.astype(dtype)
    return _lib.DeviceArray.from_host(np.asarray(x), dtype, dev)


def _wrap_out(cls, sig, noi, **kw):
    """Signal object around device results; materialised at once when KEEP_ON_DEVICE is off."""
    out = cls.from_device(sig, noi, **kw)
    if not KEEP_ON_DEVICE:
        out.signal, out.noise                               # noqa: B018 -- the attribute reads download
    return out


# ------------------------------------------------------------------ signals of the reference library itself
# A script written against opticomlib can swap single devices: `from opticomlib_amd import FIBER` and keep
# opticomlib's own PRBS / DAC / MZM / DSP around it.  Its signal objects are recognised by layout (class name,
# `.signal`, `.noise`), the sampling grid is read from THAT library's `gv` (the one the script configured), and
# the result is handed back as an object of the caller's class, so the next opticomlib device accepts it.
def _adopt(input, kind: str):
    """-> (our signal object, sampling grid, back) ; ``back(result)`` converts to the caller's class.

    ``kind``: 'optical_signal' or 'electrical_signal'.  Our own objects pass through with the global ``gv``."""
    ours = optical_signal if kind == "optical_signal" else electrical_signal
    if isinstance(input, ours):
        return input, gv, lambda out: out
    if type(input).__name__ != kind or not hasattr(input, "signal") or not hasattr(input, "noise"):
        return input, gv, lambda out: out                  # not a signal: the device raises its TypeError
    import sys
    mod = sys.modules.get(type(input).__module__)
    grid = getattr(mod, "gv", None)
    if grid is None or not all(hasattr(grid, a) for a in ("fs", "dt")):
        raise TypeError(f"cannot find the sampling grid (`gv`) of {type(input).__module__}")
    noise = input.noise if isinstance(input.noise, np.ndarray) else NULL
    mine = ours(np.asarray(input.signal), noise)
    if kind == "optical_signal" and mine.n_pol != getattr(input, "n_pol", mine.n_pol):
        mine = ours(np.asarray(input.signal), noise, n_pol=input.n_pol)

    def back(out):
        if isinstance(out, tuple):                          # (signal, H) or (z, A_z)
            return (back(out[0]),) + tuple(out[1:]) if isinstance(out[0], (optical_signal, electrical_signal)) else out
        cls = getattr(mod, type(out).__name__, None)
        if cls is None:
            return out
        res = cls(out.signal) if out.noise is NULL else cls(out.signal, out.noise)
        res.execution_time = out.execution_time
        return res
    return mine, grid, back


# ------------------------------------------------------------------ operator coefficients
_OPERATORS: "OrderedDict[tuple, np.ndarray]" = OrderedDict()
_MAX_OPERATORS = 4            # 8-16 MiB each at 2^20


def linear_operator(n, dt, alpha, beta_2, beta_3, precision=_lib.C64):
    """D~(w) = -alpha/2 + j beta_2/2 w^2 + j beta_3/6 w^3 in FFT order.

    complex64: every factor is rounded to float32 in the reference's own order
    (``devices.py:1137-1145``) so both sides start from identical coefficients.
    complex128: the float64 form of ``devices.py:2440-2442``.

    The last few operators are kept (read-only): NumPy's ``w**3`` costs ~0.1 s at 2^20 points, and a link
    simulation calls FIBER / DBP with the same fibre again and again.
    """
    key = (int(n), float(dt), float(alpha), float(beta_2), float(beta_3), int(precision))
    with _CACHE_LOCK:
        hit = _OPERATORS.get(key)
        if hit is not None:
            _OPERATORS.move_to_end(key)
            return hit
    d = _linear_operator(n, dt, alpha, beta_2, beta_3, precision)
    d.flags.writeable = False
    with _CACHE_LOCK:
        _OPERATORS[key] = d
        while len(_OPERATORS) > _MAX_OPERATORS:
            _OPERATORS.popitem(last=False)
    return d


_GRID_POWERS: "OrderedDict[tuple, tuple]" = OrderedDict()


def _grid_powers(n, dt, precision):
    """``(w**2, w**3)`` of the angular-frequency grid [rad/ps] in the plan's real type, kept for the last two grids:
    NumPy's ``w**3`` (``pow``) costs 38 ms at 2^20 points -- 12x a whole 100-step propagation -- and depends only on
    the sampling grid, so a sweep over fibre parameters pays it once."""
    key = (int(n), float(dt), int(precision))
    with _CACHE_LOCK:
        hit = _GRID_POWERS.get(key)
        if hit is not None:
            _GRID_POWERS.move_to_end(key)
            return hit
    w = np.fft.fftfreq(n, dt) * 2 * np.pi * 1e-12        # rad/ps
    if precision == _lib.C64:
        w = np.asarray(w, dtype=_F32)
    hit = (w**2, w**3)
    for a in hit:
        a.flags.writeable = False
    with _CACHE_LOCK:
        _GRID_POWERS[key] = hit
        while len(_GRID_POWERS) > 2:
            _GRID_POWERS.popitem(last=False)
    return hit


def _linear_operator(n, dt, alpha, beta_2, beta_3, precision):
    w2, w3 = _grid_powers(n, dt, precision)
    if precision == _lib.C64:
        a = np.array(alpha / 4.343, dtype=_F32)
        b2 = np.array(beta_2, dtype=_F32)
        b3 = np.array(beta_3, dtype=_F32)
        return -a / 2 + 1j / 2 * b2 * w2 + 1j / 6 * b3 * w3
    a = alpha / 4.343
    return -a / 2 + 1j / 2 * beta_2 * w2 + 1j / 6 * beta_3 * w3


def step_schedule(length, h, precision=_lib.C64):
    """Step sizes and positions of a fixed-``h`` run.

    The reference accumulates ``z`` and clamps ``h`` in float32 (``devices.py:1158-1161,
    1173,1196``), which decides the step COUNT (100 km / 0.1 km is 1001 steps); the same
    arithmetic is replayed here.  Returns ``(h[S], z[S+1])`` in the plan's real type.
    """
    rt = _F32 if precision == _lib.C64 else np.float64
    L = rt(length)
    hs = rt(h)
    if not np.isfinite(hs) or hs <= 0:
        raise ValueError(f"h must be a positive step size in km, got {h!r} (the reference never terminates for h <= 0)")
    hcur = min(hs, L)
    z = rt(0)
    hlist, zlist = [], [z]
    while z < L:
        z = rt(z + hcur)
        hlist.append(hcur)
        zlist.append(z)
        hcur = rt(min(hcur, rt(L - z)))
    return np.array(hlist, dtype=rt), np.array(zlist, dtype=rt)


# ------------------------------------------------------------------ chirp-z engine (any length)
_CHIRPS: "OrderedDict[tuple, tuple]" = OrderedDict()      # (device, n) -> (c, conj c) on the device; content-addressed, never stale


class _ChirpZ:
    """Split-step / single-transfer engine for fields of ANY length n on a power-of-two complex128 plan of
    M >= 2n - 1 points (Bluestein; algebra in csrc/chirpz.hip).  The field lives in a device array in natural
    order; all arithmetic of THIS engine is complex128 whatever the caller's precision (complex64 callers of up to 2048 samples take
    ``_fiber_chirpz_small_c64`` instead)."""

    def __init__(self, n: int, batch: int, dev: int):
        self.n, self.batch, self.dev = int(n), int(batch), int(dev)
        M = 1 << max(8, (2 * n - 2).bit_length())
        self.plan = get_plan(M, batch, _lib.C128, dev)
        with self.plan.lock:
            self._ensure_tables()
        key = (dev, n)
        with _CACHE_LOCK:
            if key not in _CHIRPS:
                _CHIRPS[key] = (_lib.chirp_device(n, False, dev), _lib.chirp_device(n, True, dev))
                while len(_CHIRPS) > 4:
                    _CHIRPS.pop(next(iter(_CHIRPS)))
            self.chirp, self.chirp_conj = _CHIRPS[key]

    def _ensure_tables(self):
        want = _tag("chirp", self.n)
        if self.plan.tag(1) != want or self.plan.tag(2) != want:
            # both convolution kernels are generated AND transformed on the device (ssfm_chirp_setup): no host FFT, no upload
            self.plan.chirp_setup(self.n)
            for slot in (0, 1):
                self.plan.set_tag(1 + slot, want)

    # the engine works in the plan's field buffer and table slots: its users hold the plan's lock for their whole sequence
    def __enter__(self):
        self.plan.lock.acquire()
        self._ensure_tables()
        return self

    def __exit__(self, *exc):
        self.plan.lock.release()
        return False

    def transfer(self, A, H, exponent=False):
        """``A <- ifft(fft(A) * H)`` (or ``* exp(H)`` with ``exponent``) for a device table ``H`` of n entries, in place."""
        self.plan.chirp_transfer(A, self.chirp, H, exponent)

    def fourier(self, A, inverse: bool):
        """``fft`` (or ``ifft``) of every row of the device array ``A`` (batch, n), in place; returns ``A``."""
        self.plan.chirp_fourier(A, self.chirp, self.chirp_conj, inverse)
        self.plan.synchronize()
        return A


def _fourier(obj, domain, shift=False):
    """``signal('w')`` / ``signal('t')`` of the reference's signal classes (``typing.py:1421-1462``): fft or ifft of
    signal and noise along the last axis -- on the device, for any length from 2 to 2^21 (complex128 chirp-z
    transform), the result stays in GPU memory.  ``shift`` applies fftshift / ifftshift (two device copies per row)."""
    if domain not in ("t", "w", "f"):
        raise ValueError("`domain` must be one of the following values ('t', 'w', 'f')")
    inverse = domain == "t"
    dev = default_device()
    raws = [obj._raw("signal")] + ([] if obj._raw("noise") is NULL else [obj._raw("noise")])
    shape = tuple(raws[0].shape)
    n = shape[-1]
    rows = 1 if len(shape) == 1 else shape[0]
    _, hi = _lib.supported_log2n(_lib.C128, direct=True)
    if n < 2 or 2 * n - 1 > (1 << hi):
        raise ValueError(f"the device transform takes 2 ... 2^{hi - 1} samples per row, got {n} (there is no CPU fallback)")
    single = all(np.dtype(a.dtype) in (np.dtype(np.complex64), np.dtype(np.float32)) for a in raws)   # NumPy >= 2 keeps single precision
    buf = _lib.DeviceArray((rows * len(raws), n), np.complex128, dev)
    row_bytes = rows * n * 16
    cp = lambda dst, src, nbytes: _lib._check(_lib.load().ssfm_device_copy(dev, _lib._VP(dst), _lib._VP(src), nbytes, 2), "ssfm_device_copy")
    for k, a in enumerate(raws):
        cp(buf.ptr + k * row_bytes, _dev_array(a, np.complex128, dev).ptr, row_bytes)
    with _ChirpZ(n, rows * len(raws), dev) as eng:
        res = eng.fourier(buf, inverse)
    outs = []
    s_ = n // 2
    for k in range(len(raws)):
        o = _lib.DeviceArray(shape, np.complex128, dev)
        for r in range(rows):
            src, dst = res.ptr + k * row_bytes + r * n * 16, o.ptr + r * n * 16
            if not shift:
                cp(dst, src, n * 16)
            elif not inverse:                                   # fftshift: out[(i + n//2) % n] = in[i]
                cp(dst + s_ * 16, src, (n - s_) * 16)
                cp(dst, src + (n - s_) * 16, s_ * 16)
            else:                                               # ifftshift: out[i] = in[(i + n//2) % n]
                cp(dst, src + s_ * 16, (n - s_) * 16)
                cp(dst + (n - s_) * 16, src, s_ * 16)
        outs.append(o.astype(np.complex64) if single else o)
    kw = {"n_pol": obj.n_pol} if isinstance(obj, optical_signal) else {}
    return _wrap_out(type(obj), outs[0], outs[1] if len(outs) > 1 else NULL, **kw)


def _fiber_chirpz(A_dev, shape, dt, length, alpha, beta_2, beta_3, gamma, phi_max, h, return_steps, prec, dev, bar):
    """FIBER for a length that is not a power of two (same semantics as the fused engine: float32 coefficients
    and step arithmetic in complex64 mode, reference devices.py:1137-1161, 1172-1196)."""
    n = shape[-1]
    batch = 1 if len(shape) == 1 else shape[0]
    rt = _F32 if prec == _lib.C64 else np.float64
    if prec == _lib.C64 and n <= 2048 and not return_steps and bar is None and os.environ.get("SSFM_CHIRP_LOOP", "c") != "python" \
            and os.environ.get("SSFM_CHIRP_SMALL", "1") != "0" and batch <= 16:
        got = _fiber_chirpz_small_c64(A_dev, shape, dt, length, alpha, beta_2, beta_3, gamma, phi_max, h, dev)
        if got is not None:
            return got + ("chirp_small_c64",)
    if prec == _lib.C64 and 2048 < n <= 65536 and not return_steps and bar is None and os.environ.get("SSFM_CHIRP_LOOP", "c") != "python" \
            and os.environ.get("SSFM_MEDIUM", "1") != "0":
        got = _fiber_chirpz_medium_c64(A_dev, shape, dt, length, alpha, beta_2, beta_3, gamma, phi_max, h, dev)
        if got is not None:
            return got + ("chirp_medium_c64",)
    with _ChirpZ(n, batch, dev) as eng:
        return _fiber_chirpz_locked(eng, A_dev, shape, dt, length, alpha, beta_2, beta_3, gamma, phi_max, h, return_steps, prec, dev, bar) + ("chirp_line_c128",)


_CHIRPS32: "OrderedDict[tuple, object]" = OrderedDict()

# Accuracy margin of the one-launch complex64 chirp-z line (2048 < n <= 65536; profiles/r05_chirp_margin.txt).  A step of that line is four
# padded complex64 transforms of 2-4 x the field's length where the reference's pocketfft makes two of the length itself: measured over 242 random
# fibres, max|A - A_float64|/peak <= 7.5e-7 x steps^0.75.  The reference's own complex64 run sits up to the tolerance itself from the float64
# solution for such lengths, so the line may use HALF of the stated bound -- `accuracy.tol(steps)`, ONE bound continuous in the number of steps
# (2e-5 up to 100 steps, the log-log line to 3e-4 at 1000, proportional beyond).  Where the law exceeds half the bound the run takes the complex128
# line (four launches per step, 1e-13 from float64): 27 ... 1031 steps, DERIVED from the law and the bound (`accuracy.c64_line_window`).  Round 5's
# window, 32 ... 100, was fitted to the step at 101 steps of the tolerance the tests then used (2e-5 up to 100 steps, 3e-4 from 101): a 101-step
# run is 2.4e-5 from the float64 solution by the line's own law.  `precision="complex128"` always takes the complex128 line.
_c64_line_has_margin = accuracy.c64_line_has_margin
_C64_LINE_NO_MARGIN = accuracy.c64_line_window()          # (first, last) step count without margin: (27, 1031)


def _fiber_chirpz_small_c64(A_dev, shape, dt, length, alpha, beta_2, beta_3, gamma, phi_max, h, dev):
    """``precision="complex64"`` at up to 2048 samples that are not a power of two: the whole run in ONE launch on a complex64 line of M >= 2n - 1 <= 4096 points
    (k_small_chirp / k_small_chirp_adapt instantiated for float; round 4).  The reference itself transforms such a length in single precision (pocketfft's Bluestein,
    devices.py:1178-1180), so this is its arithmetic class: 2e-6 ... 7e-6 from the float64 solution after 100 steps (profiles/r04_chirpz_c64_error.txt) where the
    complex128 line of the general path sits at 1e-13 -- and a step takes half the time (one CU does the four line transforms of a step; float64 was its bound).
    ``precision="complex128"`` keeps the complex128 line.  None: the plan has no such engine, or its rows' workgroups did not meet: the general path runs."""
    n = shape[-1]
    batch = 1 if len(shape) == 1 else shape[0]
    M = 1 << max(8, (2 * n - 2).bit_length())
    plan = get_plan(M, batch, _lib.C64, dev)
    L = _F32(length)
    if not float(L) > 0:
        A = A_dev if A_dev.dtype == np.complex64 else A_dev.astype(np.complex64)
        return (A.copy() if A is A_dev else A), [_F32(0)], None
    with plan.lock:
        key = (dev, n)
        with _CACHE_LOCK:
            if key not in _CHIRPS32:
                _CHIRPS32[key] = _lib.chirp_device(n, False, dev).astype(np.complex64)
                while len(_CHIRPS32) > 4:
                    _CHIRPS32.pop(next(iter(_CHIRPS32)))
            chirp = _CHIRPS32[key]
        A = A_dev.astype(np.complex64) if A_dev.dtype != np.complex64 else A_dev.copy()          # the caller's array is never modified
        A.shape = (batch, n)
        Dt = _lib.DeviceArray.from_host(np.asarray(linear_operator(n, dt, alpha, beta_2, beta_3, _lib.C64), dtype=np.complex64), np.complex64, dev)
        g = float(_F32(gamma))
        if h is None:
            b2, b3 = _F32(beta_2), _F32(beta_3)
            if bool((b2 == 0 and b3 == 0) or _F32(gamma) == 0):                  # one step of the whole length (reference devices.py:1163-1170)
                if not plan.chirp_propagate_c64(A, chirp, Dt, g, np.array([float(L)])):
                    return None
                zs = [_F32(0), L]
            else:
                zs, z0, max_steps = [_F32(0)], _F32(0), 1 << 17
                while True:
                    got = plan.chirp_propagate_c64(A, chirp, Dt, g, None, length=float(_F32(L - z0)), phi_max=float(_F32(phi_max)), max_steps=max_steps)
                    if got is None:
                        return None if z0 == 0 else _raise_chirp_midway()
                    steps, z = got
                    zs += [_F32(z0 + _F32(v)) for v in z[1:]]
                    if steps < max_steps or not (zs[-1] < L):
                        break
                    z0 = zs[-1]
        else:
            hs, z_all = step_schedule(length, h, _lib.C64)
            if not plan.chirp_propagate_c64(A, chirp, Dt, g, np.asarray(hs, dtype=np.float64)):
                return None
            zs = list(z_all)
        plan.synchronize()
        return A, zs, None


def _fiber_chirpz_medium_c64(A_dev, shape, dt, length, alpha, beta_2, beta_3, gamma, phi_max, h, dev):
    """``precision="complex64"``, 2048 < n <= 65536 samples that are not a power of two: the whole run in one launch on one XCD on a complex64 line of
    M >= 2n - 1 points, 2^13 ... 2^17 (k_medium_chirp / k_medium_chirp_adapt: four passes per step instead of five / seven launches; round 4).  The arithmetic
    class of ``_fiber_chirpz_small_c64``.  None: a plan whose rows do not fit the one-XCD engine (2^17 points in all), more than four step sizes, or a launch
    whose workgroups did not meet: the general path runs."""
    n = shape[-1]
    batch = 1 if len(shape) == 1 else shape[0]
    M = 1 << (2 * n - 2).bit_length()
    L = _F32(length)
    if M * batch > (1 << 17) or not float(L) > 0:
        return None
    adaptive = False
    if h is None:
        b2, b3 = _F32(beta_2), _F32(beta_3)
        adaptive = not bool((b2 == 0 and b3 == 0) or _F32(gamma) == 0)
        hs, zs = np.array([float(L)]), [_F32(0), L]                             # one step of the whole length (reference devices.py:1163-1170)
        if adaptive and os.environ.get("SSFM_MEDIUM_ADAPT", "1") == "0":
            return None
    else:
        hs, z_all = step_schedule(length, h, _lib.C64)
        hs, zs = np.asarray(hs, dtype=np.float64), list(z_all)
    if len(set(hs.tolist())) > 4 or not _c64_line_has_margin(len(hs)):
        return None
    plan = get_plan(M, batch, _lib.C64, dev)
    with plan.lock:
        key = (dev, n)
        with _CACHE_LOCK:
            if key not in _CHIRPS32:
                _CHIRPS32[key] = _lib.chirp_device(n, False, dev).astype(np.complex64)
                while len(_CHIRPS32) > 4:
                    _CHIRPS32.pop(next(iter(_CHIRPS32)))
            chirp = _CHIRPS32[key]
        A = A_dev.astype(np.complex64) if A_dev.dtype != np.complex64 else A_dev.copy()          # the caller's array is never modified
        A.shape = (batch, n)
        Dt = _lib.DeviceArray.from_host(np.asarray(linear_operator(n, dt, alpha, beta_2, beta_3, _lib.C64), dtype=np.complex64), np.complex64, dev)
        g = float(_F32(gamma))
        if not adaptive:
            if not plan.chirp_propagate_c64(A, chirp, Dt, g, hs):
                return None
            return A, zs, None
        # adaptive: the step count is the run's own.  One step on a copy gives the first step size; the run's length in steps follows from it
        # (h grows as the power falls: steps ~ L_eff / h0) -- a run that will end inside the window without margin goes to the complex128 line at once
        probe = A.copy()
        got = plan.chirp_propagate_c64(probe, chirp, Dt, g, None, length=float(L), phi_max=float(_F32(phi_max)), max_steps=1)
        if got is None:
            return None
        h0 = float(got[1][1]) if got[0] >= 1 else float(L)
        a_lin = float(_F32(alpha / 4.343))
        x = a_lin * float(L)
        l_eff = float(L) if abs(x) < 1e-6 else float(L) * (1.0 - np.exp(-x)) / x
        est = l_eff / h0 if h0 > 0 else 1.0
        if 0.85 * _C64_LINE_NO_MARGIN[0] <= est <= 1.15 * _C64_LINE_NO_MARGIN[1]:
            return None
        zs, z0, max_steps = [_F32(0)], _F32(0), 1 << 17
        while True:
            got = plan.chirp_propagate_c64(A, chirp, Dt, g, None, length=float(_F32(L - z0)), phi_max=float(_F32(phi_max)), max_steps=max_steps)
            if got is None:
                return None if z0 == 0 else _raise_chirp_midway()
            steps, z = got
            zs += [_F32(z0 + _F32(v)) for v in z[1:]]
            if steps < max_steps or not (zs[-1] < L):
                break
            z0 = zs[-1]
        if not _c64_line_has_margin(len(zs) - 1):              # (the estimate was off: the run is repeated on the complex128 line)
            return None
        return A, zs, None


def _raise_chirp_midway():
    raise _lib.SsfmError("the one-launch chirp-z engine lost its workgroups in the middle of a run of more than 131072 steps")


def _fiber_chirpz_locked(eng, A_dev, shape, dt, length, alpha, beta_2, beta_3, gamma, phi_max, h, return_steps, prec, dev, bar):
    n = shape[-1]
    batch = 1 if len(shape) == 1 else shape[0]
    rt = _F32 if prec == _lib.C64 else np.float64
    A = A_dev if A_dev.dtype == np.complex128 else A_dev.astype(np.complex128)
    if A is A_dev:
        A = A.copy()                                            # the caller's array is never modified
    A.shape = (batch, n)
    P = _lib.DeviceArray((batch, n), np.float64, dev)
    Dt = _lib.DeviceArray.from_host(np.asarray(linear_operator(n, dt, alpha, beta_2, beta_3, prec), dtype=np.complex128), np.complex128, dev)
    g = float(rt(gamma))
    L = rt(length)
    zs, snaps = [rt(0)], ([A.copy()] if return_steps else None)
    steps = 0
    if not return_steps and bar is None and os.environ.get("SSFM_CHIRP_LOOP", "c") != "python":
        # the whole run from C (ssfm_chirp_propagate): no host call per kernel, the adaptive step rule on the device
        if h is None:
            b2, b3 = rt(beta_2), rt(beta_3)
            if bool((b2 == 0 and b3 == 0) or rt(gamma) == 0):                  # one step of the whole length (reference devices.py:1163-1170)
                eng.plan.chirp_propagate(A, P, eng.chirp, Dt, g, np.array([float(L)]), f32=(prec == _lib.C64))
                zs = [rt(0), L]
            else:
                if not float(L) > 0:
                    return A, zs, snaps
                max_steps, z0 = 1 << 17, rt(0)
                while True:                                   # (one call unless a run needs more than 131072 steps: then it goes on from where it is)
                    steps, z = eng.plan.chirp_propagate(A, P, eng.chirp, Dt, g, None, length=float(rt(L - z0)), phi_max=float(rt(phi_max)),
                                                        f32=(prec == _lib.C64), max_steps=max_steps)
                    zs += [rt(z0 + rt(v)) for v in z[1:]]
                    if steps < max_steps or not (zs[-1] < L):
                        break
                    z0 = zs[-1]
        else:
            hs, z_all = step_schedule(length, h, prec)
            eng.plan.chirp_propagate(A, P, eng.chirp, Dt, g, np.asarray(hs, dtype=np.float64), f32=(prec == _lib.C64))
            zs = list(z_all)
        return A, zs, snaps
    # A caller that wants the field after every step (return_steps), a progress bar, or SSFM_CHIRP_LOOP=python: the same C entry point a step at a time
    # (ssfm_chirp_propagate with one step size, or -- adaptive -- with max_steps = 1 over what is left of the length: the step rule is evaluated on the
    # device in the caller's arithmetic either way, so the schedule is the whole run's bit for bit)
    def after_step(hk):
        if return_steps:
            snaps.append(A.copy())
        if bar is not None:
            bar.update(min(100.0 * float(hk) / float(L), max(0.0, 100.0 - bar.n)))
            bar.set_postfix(FFTs=2 * steps)
    if h is None:
        b2, b3 = rt(beta_2), rt(beta_3)
        single = bool((b2 == 0 and b3 == 0) or rt(gamma) == 0)
        z = rt(0)
        while z < L:
            if single:
                eng.plan.chirp_propagate(A, P, eng.chirp, Dt, g, np.array([float(L)]), f32=(prec == _lib.C64))
                hk = L
            else:
                _, zz = eng.plan.chirp_propagate(A, P, eng.chirp, Dt, g, None, length=float(rt(L - z)), phi_max=float(rt(phi_max)), f32=(prec == _lib.C64), max_steps=1)
                hk = rt(zz[1])
            z = rt(z + hk)
            steps += 1
            zs.append(z)
            after_step(hk)
    else:
        hs, z_all = step_schedule(length, h, prec)
        for k, hk in enumerate(hs):
            eng.plan.chirp_propagate(A, P, eng.chirp, Dt, g, np.array([float(hk)]), f32=(prec == _lib.C64))
            steps += 1
            after_step(hk)
        zs = list(z_all)
    eng.plan.synchronize()
    return A, zs, snaps


# ------------------------------------------------------------------ FIBER / DBP
def FIBER(input: optical_signal,
          length: float,
          alpha: float = 0.0,
          beta_2: float = 0.0,
          beta_3: float = 0.0,
          gamma: float = 0.0,
          phi_max: float = 0.01,
          h: float = None,
          show_progress: bool = False,
          return_steps: bool = False,
          *,
          precision="complex64",
          device=None,
          every: int = None,
          z_list=None):
    """Optical fibre: symmetric split-step Fourier solution of the scalar NLSE per polarisation.

    Parameters as the reference (``devices.py:1038-1083``): ``length`` [km], ``alpha`` [dB/km],
    ``beta_2`` [ps^2/km], ``beta_3`` [ps^3/km], ``gamma`` [1/(W km)], ``phi_max`` [rad] bound of
    the nonlinear phase per adaptive step, ``h`` fixed step [km] or ``None`` for adaptive.
    Returns an :class:`optical_signal` (``noise`` folded into ``signal``, ``noise = NULL``) or,
    with ``return_steps``, ``(z float64 (S+1,), A_z (S+1, [2,] N))``.

    Extensions (keyword-only, defaults keep the reference's behaviour): ``precision``
    ``"complex64"`` (the reference's arithmetic) or ``"complex128"``; ``device`` index; ``every``
    (with ``return_steps`` and a fixed ``h`` on a power-of-two length): keep the field after every
    ``every``-th step only (plus the input and the last step) -- the run stays on the fused engine and the
    snapshots travel to page-locked host memory beside it (``ssfm_propagate_fixed_capture``); the reference
    keeps every step, 16 GiB for the 1000-step run of a 2^20-sample dual-polarisation field.  With the adaptive step
    (``h=None``, the reference's default and what its own consumer of ``return_steps`` uses, ``devices.py:2342``) ``every`` and
    ``z_list`` (round 6) keep the run's own engine as well (``ssfm_adaptive_set_capture``): ``z_list`` = positions [km] -- for each
    the field after the first step that reaches it (the input for z <= 0, the end field beyond the length).  The run is
    made twice: once for its z log (an adaptive run repeats bit for bit), once with the capture at the steps that follow from it.
    """
    t0 = time.time()
    input, grid, back = _adopt(input, "optical_signal")
    if not isinstance(input, optical_signal):
        raise TypeError("`input` must be of type 'optical_signal'.")
    prec = _precision_code(precision)
    plan_dtype = np.complex64 if prec == _lib.C64 else np.complex128
    rt = _F32 if prec == _lib.C64 else np.float64

    raw_s, raw_n = input._raw("signal"), input._raw("noise")
    dev = default_device() if device is None else int(device)
    A = A_dev = None
    if _on_device(raw_s) or _on_device(raw_n):
        # signal + noise in their common type, then the cast -- NumPy's own order (devices.py:1147)
        if raw_n is NULL:
            A_dev = _dev_array(raw_s, plan_dtype, dev)
        else:
            common = np.result_type(raw_s.dtype, raw_n.dtype, np.complex64)
            total = _dev_array(raw_s, common, dev) + _dev_array(raw_n, common, dev)
            A_dev = total if total.dtype == plan_dtype else total.astype(plan_dtype)
        shape = tuple(A_dev.shape)
    else:
        A = np.asarray(input.to_numpy())
        shape = A.shape
    n = shape[-1]
    batch = 1 if len(shape) == 1 else shape[0]
    _check_size(n, prec)
    if A is not None and _is_fast_size(n, prec):
        if A.dtype == np.complex128 and plan_dtype == np.complex64 and A.flags.c_contiguous:
            # the usual case (NumPy signals are complex128): upload as it lies and round to complex64 on the
            # device -- the same round-to-nearest cast, without a 32 -> 16 MiB conversion pass on the host
            A_dev = _lib.DeviceArray.from_host(A, np.complex128, dev).astype(np.complex64)
        else:
            A = np.ascontiguousarray(A, dtype=plan_dtype)

    if not _is_fast_size(n, prec):
        # any other length: chirp-z path on a power-of-two plan of >= 2n - 1 points (complex128 arithmetic; complex64 for complex64 callers of up to 2048 samples)
        bar = None
        if show_progress:
            try:
                from tqdm.auto import tqdm
                bar = tqdm(total=100, desc="Propagating", bar_format="{l_bar}{bar}|[{elapsed}{postfix}]", postfix={"FFTs": 0})
            except ImportError:
                bar = None
        if A_dev is None:
            A_dev = _lib.DeviceArray.from_host(np.ascontiguousarray(A, dtype=np.complex128), np.complex128, dev)
        out, zs, snaps, engine = _fiber_chirpz(A_dev, shape, float(grid.dt), length, alpha, beta_2, beta_3, gamma, phi_max, h, return_steps, prec, dev, bar)
        if bar is not None:
            bar.close()
        if return_steps:
            keep = _every_index(len(snaps) - 1, every)
            if z_list is not None:                      # the first step that reaches each position (0: the input; the last: the end field)
                zs64 = np.asarray(zs, dtype=np.float64)
                keep = [int(min(np.searchsorted(zs64, v, side="left"), len(snaps) - 1)) for v in np.asarray(z_list, dtype=np.float64).ravel()]
            A_z = np.stack([snaps[k].to_host() for k in keep]).astype(plan_dtype).reshape((len(keep),) + tuple(shape))
            return np.asarray(zs, dtype=np.float64)[keep], A_z
        res = out if out.dtype == plan_dtype else out.astype(plan_dtype)
        res.shape = tuple(shape)
        output = _wrap_out(optical_signal, res, NULL)
        output.execution_time = time.time() - t0
        output.engine = engine                                # which chirp-z line ran: "chirp_small_c64", "chirp_medium_c64" (one launch, complex64) or "chirp_line_c128"
        return back(output)

    L = rt(length)
    plan = get_plan(n, batch, prec, dev)
    with plan.lock:
        return back(_fiber_on_plan(plan, A, A_dev, shape, float(grid.dt), L, length, alpha, beta_2, beta_3, gamma, phi_max, h,
                                   show_progress, return_steps, prec, plan_dtype, rt, dev, t0, every, z_list))


def _every_index(steps: int, every):
    """Indices of the snapshots ``every`` keeps of a run of ``steps`` steps: 0 (the input), every ``every``-th step, the last one."""
    if every is None:
        return list(range(steps + 1))
    every = int(every)
    if every < 1:
        raise ValueError(f"every = {every}")
    return list(range(0, steps, every)) + [steps]


def _adaptive_capture(plan, gamma, length, phi_max, every, z_list, batch, n):
    """``FIBER(h=None, return_steps=True, every=k | z_list=[...])`` on a power-of-two plan: the run once for its z log, the input restored, the run again
    with the capture at the step numbers that follow (``Plan.propagate_adaptive_capture``).  Returns (z of the kept snapshots, snapshots)."""
    x0 = _lib.DeviceArray((batch, n), plan.cdtype, plan.device)
    plan.get_field_device(x0.ptr)
    first = plan.get_field().reshape(1, batch, n)
    # (the first pass on the engine the capture will run on: a capture that never fires -- the one-launch engines of the small plans find their
    # step sizes in another order of operations, and their z log differs from the launch-per-pass engines' in the last bits)
    steps, z, _, _ = plan.propagate_adaptive_capture(gamma, length, phi_max, steps=np.array([1 << 40], dtype=np.int64))
    if z_list is not None:
        zl = np.asarray(z_list, dtype=np.float64).ravel()
        keep = [int(min(np.searchsorted(z, rt_z, side="left"), steps)) for rt_z in zl]          # the first step with z_step >= z (0: the input; steps: the end)
    else:
        keep = _every_index(steps, every)
    want = sorted({k for k in keep if 0 < k < steps})
    got = {0: first[0]}
    if want:
        plan.set_field_device(x0.ptr)
        steps2, z2, taken, fields = plan.propagate_adaptive_capture(gamma, length, phi_max, steps=np.asarray(want, dtype=np.int64))
        if steps2 != steps or not np.array_equal(z2, z) or list(taken) != want:
            raise _lib.SsfmError(f"the adaptive run did not repeat ({steps} / {steps2} steps, {len(taken)} of {len(want)} snapshots)")
        for k, f in zip(want, fields):
            got[k] = f
    got[steps] = plan.get_field().reshape(batch, n) if steps else first[0]
    return np.asarray(z)[keep], np.stack([got[k] for k in keep])


def _fiber_on_plan(plan, A, A_dev, shape, dt, L, length, alpha, beta_2, beta_3, gamma, phi_max, h, show_progress, return_steps,
                   prec, plan_dtype, rt, dev, t0, every=None, z_list=None):
    """The fused two-kernel engine behind ``FIBER`` (the caller holds the plan's lock for the whole sequence)."""
    n = shape[-1]
    batch = 1 if len(shape) == 1 else shape[0]
    op_tag = _tag("fibre", dt, float(alpha), float(beta_2), float(beta_3))
    if plan.tag(0) != op_tag:                             # D~ is O(N) host work + an upload: reuse it while the plan still holds it
        plan.set_linear_operator(linear_operator(n, dt, alpha, beta_2, beta_3, prec))
        plan.set_tag(0, op_tag)
    if A_dev is not None:
        plan.set_field_device(A_dev.ptr)
    else:
        plan.set_field(A)

    bar = None
    if show_progress:
        try:
            from tqdm.auto import tqdm
            bar = tqdm(total=100, desc="Propagating", bar_format="{l_bar}{bar}|[{elapsed}{postfix}]", postfix={"FFTs": 0})
        except ImportError:
            bar = None

    if h is None:
        b2, b3, g = rt(beta_2), rt(beta_3), rt(gamma)
        single = bool((b2 == 0 and b3 == 0) or g == 0)
        if return_steps and (every is not None or z_list is not None) and not single and n <= (1 << _lib.DIRECT_LOG2_MAX) and float(L) > 0:
            z, snaps = _adaptive_capture(plan, gamma, length, phi_max, every, z_list, batch, n)
            steps, every, z_list = len(z) - 1, None, None                         # (thinned already)
        else:
            steps, z, snaps = plan.propagate_adaptive(gamma, length, phi_max, single, snapshots=return_steps)
            if return_steps and z_list is not None:                              # (a single step, a split plan: every snapshot was taken; picked here)
                zl = np.asarray(z_list, dtype=np.float64).ravel()
                keep = [int(min(np.searchsorted(np.asarray(z, dtype=np.float64), v, side="left"), len(z) - 1)) for v in zl]
                snaps, z, every = snaps[keep], np.asarray(z)[keep], None
    else:
        hs, z = step_schedule(length, h, prec)
        steps = hs.size
        snaps = None
        if return_steps and every is not None and steps and n > (1 << _lib.DIRECT_LOG2_MAX):
            # a split plan (more than 2^22 samples per row) has no capture beside the run: `every` steps per call, the field read back in between
            k = _every_index(steps, every)
            fields = [plan.get_field()]
            for a0, a1 in zip(k[:-1], k[1:]):
                plan.propagate_fixed(gamma, hs[a0:a1])
                fields.append(plan.get_field())
            snaps, z, every = np.stack(fields), np.asarray(z)[k], None
        elif return_steps and every is not None and steps:
            cap = plan.propagate_fixed_capture(gamma, hs, every=every)
            snaps, z, every = cap["fields"], np.asarray(z)[cap["steps"]], None              # (thinned already)
        elif return_steps:
            snaps = plan.propagate_fixed(gamma, hs, snapshots=True) if steps else plan.get_field().reshape(1, batch, n)
        elif bar is not None and steps:
            done = 0
            for chunk in np.array_split(hs, min(20, steps)):      # progress needs sync points
                plan.propagate_fixed(gamma, chunk)
                plan.synchronize()
                done += chunk.size
                bar.update(min(100.0 * float(chunk.sum()) / float(L), max(0.0, 100.0 - bar.n)))
                bar.set_postfix(FFTs=2 * done)
        elif steps:
            plan.propagate_fixed(gamma, hs)
    if bar is not None:
        if h is None or return_steps:
            bar.update(100)
            bar.set_postfix(FFTs=2 * int(steps))
        bar.close()

    if return_steps:
        if every is not None and snaps.shape[0] == len(z):         # (the adaptive run, or a run of no steps: every snapshot was taken; thinned here)
            keep = _every_index(snaps.shape[0] - 1, every)
            snaps, z = snaps[keep], np.asarray(z)[keep]
        A_z = snaps.reshape((snaps.shape[0],) + shape)
        return np.asarray(z, dtype=np.float64), A_z
    if KEEP_ON_DEVICE:
        out = _lib.DeviceArray(shape, plan_dtype, dev)
        plan.get_field_device(out.ptr)
        output = optical_signal.from_device(out)
    else:
        output = optical_signal(plan.get_field().reshape(shape))
    output.execution_time = time.time() - t0
    # beside the reference's execution_time: which engine ran ("two_kernel", "medium", "adaptive_fused" ..., _lib.ENGINES) and whether a one-launch
    # engine had to give way to its fallback -- a shared GPU or a profiler turns the fast engines off without any other sign
    info = plan.last_run_info()
    output.engine = info["engine"] + (" (fell back)" if info["fell_back"] else "")
    return output


def DBP(input: optical_signal,
        length: float,
        alpha: float = 0.0,
        beta_2: float = 0.0,
        beta_3: float = 0.0,
        gamma: float = 0.0,
        phi_max: float = 0.01,
        h: float = None,
        show_progress: bool = False,
        return_steps: bool = False,
        *,
        precision="complex64",
        device=None,
        every: int = None,
        z_list=None):
    """Digital back-propagation = ``FIBER`` with every operator negated (``devices.py:1280-1283``)."""
    return FIBER(input, length=length, alpha=-alpha, beta_2=-beta_2, beta_3=-beta_3, gamma=-gamma,
                 phi_max=phi_max, h=h, show_progress=show_progress, return_steps=return_steps,
                 precision=precision, device=device, every=every, z_list=z_list)


# ------------------------------------------------------------------ DM
def DM(input: optical_signal, D: float, retH: bool = False, *, device=None):
    """Dispersive medium: one linear step ``H(w) = exp(+j w^2 D/2)``, ``D = beta_2 z`` [ps^2]
    (``devices.py:1019-1035``).  complex128; signal and noise are filtered separately and stay
    separate.  With ``retH`` also returns ``fftshift(H)``."""
    t0 = time.time()
    input, grid, back = _adopt(input, "optical_signal")
    if not isinstance(input, optical_signal):
        raise TypeError("`input` must be of type 'optical_signal'.")
    D = D * 1e-12**2          # ps^2 -> s^2 (devices.py:1025); H(w) is generated on the device

    raw_s, raw_n = input._raw("signal"), input._raw("noise")
    dev = default_device() if device is None else int(device)
    shape = tuple(raw_s.shape)
    n = shape[-1]
    rows = 1 if len(shape) == 1 else shape[0]
    _check_size(n, _lib.C128)
    has_noise = raw_n is not NULL
    if not _is_fast_size(n, _lib.C128):
        # any other length: chirp-z path with H = exp(+j w^2 D / 2) formed on the host in the reference's own
        # float64 expression (devices.py:1025-1027)
        w = np.fft.fftfreq(n, float(grid.dt)) * 2 * np.pi           # typing.py:1641
        phase = 1j * w ** 2 * D / 2                                 # the exponential itself is taken on the device
        Hd = _lib.DeviceArray.from_host(phase, np.complex128, dev)
        nrow = rows * (2 if has_noise else 1)
        buf = _lib.DeviceArray((nrow, n), np.complex128, dev)
        for k, a in enumerate([raw_s, raw_n] if has_noise else [raw_s]):
            d = _dev_array(a, np.complex128, dev)
            _lib._check(_lib.load().ssfm_device_copy(dev, _lib._VP(buf.ptr + k * rows * n * 16), _lib._VP(d.ptr), rows * n * 16, 2), "ssfm_device_copy")
        with _ChirpZ(n, nrow, dev) as eng:
            eng.transfer(buf, Hd, exponent=True)
            eng.plan.synchronize()
        outs = []
        for k in range(2 if has_noise else 1):
            o = _lib.DeviceArray(shape, np.complex128, dev)
            _lib._check(_lib.load().ssfm_device_copy(dev, _lib._VP(o.ptr), _lib._VP(buf.ptr + k * rows * n * 16), rows * n * 16, 2), "ssfm_device_copy")
            outs.append(o)
        output = _wrap_out(optical_signal, outs[0], outs[1] if has_noise else NULL, n_pol=input.n_pol)
        output.execution_time = time.time() - t0
        if retH:
            return back(output), np.fft.fftshift(np.exp(phase))
        return back(output)
    plan = get_plan(n, rows * (2 if has_noise else 1), _lib.C128, dev)
    # signal rows, then noise rows, straight into the plan's field buffer
    row_bytes = rows * n * 16
    with plan.lock:
        for k, a in enumerate([raw_s, raw_n] if has_noise else [raw_s]):
            if _on_device(a):
                d = _dev_array(a, np.complex128, dev)
                plan.copy_into_field(k * row_bytes, d.ptr, row_bytes, on_device=True)
            else:
                h = np.ascontiguousarray(a, dtype=np.complex128)
                plan.copy_into_field(k * row_bytes, h.ctypes.data, row_bytes, on_device=False)
        H = plan.apply_dispersion(float(grid.dt), D, want_H=retH)
        outs = []
        for k in range(2 if has_noise else 1):
            o = _lib.DeviceArray(shape, np.complex128, dev)
            plan.copy_from_field(k * row_bytes, o.ptr, row_bytes)
            outs.append(o)
    output = _wrap_out(optical_signal, outs[0], outs[1] if has_noise else NULL, n_pol=input.n_pol)
    output.execution_time = time.time() - t0
    if retH:
        return back(output), np.fft.fftshift(H)
    return back(output)


# ------------------------------------------------------------------ LPF / BPF
def _warn_narrow(cutoff_hz, fs, what):
    """The time-parallel scan behind the zero-phase filters propagates filter states over thousands of samples with
    powers of the state matrix; for a cutoff far below the sampling rate those powers are ill-conditioned and the
    result drifts away from SciPy's sample-by-sample recursion: about 5e-20 (fs / cutoff)^3 relative -- 6e-14 at
    cutoff = fs / 100, 1e-11 at fs / 500, 5e-8 at fs / 10000 (tests/diag/filter_conditioning.py)."""
    if cutoff_hz < 2e-3 * fs:
        import warnings
        warnings.warn(f"{what}: cutoff {cutoff_hz:.3g} Hz is below fs/500 ({fs / 500:.3g} Hz); the device filter then differs from "
                      f"scipy.signal.sosfiltfilt by about {5e-20 * (fs / cutoff_hz) ** 3:.1e} (relative)", RuntimeWarning, stacklevel=3)


@functools.lru_cache(maxsize=64)
def _bessel_design(n, cutoff_hz, fs):
    from scipy import signal as sg
    sos = sg.bessel(N=n, Wn=cutoff_hz, btype="low", fs=fs, output="sos", norm="mag")
    zi = sg.sosfilt_zi(sos)
    sos.setflags(write=False)
    zi.setflags(write=False)
    return sos, zi


def _bessel_sos(n, cutoff_hz, fs):
    """Filter DESIGN (O(order) host work, identical call to the reference's): Bessel low-pass as
    second-order sections, magnitude-normalised, plus the steady-state initial conditions.  SciPy
    needs a millisecond for it -- 20 to 50 times the filter kernels of a call -- so the (read-only)
    result of the last 64 distinct (order, cutoff, fs) is kept."""
    return _bessel_design(int(n) if isinstance(n, (int, np.integer)) else n, float(cutoff_hz), float(fs))


def LPF(input, BW: float, n: int = 4, fs: float = None, retH: bool = False, *, device=None):
    """Electrical low-pass filter: ``n``-th order Bessel, zero phase (``sosfiltfilt``), cutoff ``BW``
    [Hz]; reference ``devices.py:1286-1375``.  ``input``: 1-D array or :class:`electrical_signal`;
    signal and noise are filtered separately and the real part is kept.  With ``retH`` also returns
    ``fftshift`` of the frequency response over ``input.size`` points."""
    t0 = time.time()
    input, grid, back = _adopt(input, "electrical_signal")
    if not isinstance(input, electrical_signal):
        input = electrical_signal(input)
    if input.ndim != 1:
        raise ValueError("`input` must be a 1D-array.")
    if not fs:
        fs = grid.fs
    sos, zi = _bessel_sos(n, BW, fs)
    _warn_narrow(BW, fs, "LPF")
    dev = default_device() if device is None else int(device)
    # real coefficients: Re(filter(x)) == filter(Re(x)), so only the real channel is computed
    # (signal and noise go to the device as they lie: no stacked host copy)
    has_noise = input._raw("noise") is not NULL
    res = []
    for a in ([input._raw("signal"), input._raw("noise")] if has_noise else [input._raw("signal")]):
        if _on_device(a) and a.dtype != np.float64:
            a = _lib.real_device(a if a.dtype == np.complex128 else a.astype(np.complex128))     # complex on the device: its real part, there
        x = a if _on_device(a) else _lib.DeviceArray.from_host(np.real(a), np.float64, dev)
        y = _lib.DeviceArray(x.shape, np.float64, dev)
        _lib.sosfiltfilt_device(sos, zi, x.ptr, y.ptr, x.shape[-1], 1, False, dev)
        res.append(y)
    output = _wrap_out(electrical_signal, res[0], res[1] if has_noise else NULL)
    output.execution_time = time.time() - t0
    if retH:
        from scipy import signal as sg
        _, H = sg.sosfreqz(sos, worN=input.size, fs=fs, whole=True)
        return back(output), np.fft.fftshift(H)
    return back(output)


def _bpf_on_grid(x, BW, grid, device):
    """BPF of one of OUR signals on another library's sampling grid (EDFA of an adopted input)."""
    if grid is gv:
        return BPF(x, BW, device=device)
    saved = (gv.fs, gv.dt)
    gv.fs, gv.dt = grid.fs, grid.dt
    try:
        return BPF(x, BW, device=device)
    finally:
        gv.fs, gv.dt = saved


def BPF(input: optical_signal, BW: float, n: int = 4, *, device=None):
    """Optical band-pass filter: Bessel low-pass of cutoff ``BW/2`` on the complex envelope, zero phase;
    reference ``devices.py:788-826``.  Signal and noise are filtered separately."""
    t0 = time.time()
    input, grid, back = _adopt(input, "optical_signal")
    if not isinstance(input, optical_signal):
        raise TypeError("`input` must be of type (optical_signal).")
    sos, zi = _bessel_sos(n, BW / 2, grid.fs)
    _warn_narrow(BW / 2, grid.fs, "BPF")
    dev = default_device() if device is None else int(device)
    has_noise = input._raw("noise") is not NULL
    res = []
    for a in ([input._raw("signal"), input._raw("noise")] if has_noise else [input._raw("signal")]):
        cplx = np.dtype(a.dtype).kind == "c"                # a real envelope (a CW laser) stays real, as in SciPy
        x = _dev_array(a, np.complex128 if cplx else np.float64, dev)
        y = _lib.DeviceArray(x.shape, x.dtype, dev)
        _lib.sosfiltfilt_device(sos, zi, x.ptr, y.ptr, x.shape[-1], x.size // x.shape[-1], cplx, dev)
        res.append(y)
    output = _wrap_out(optical_signal, res[0], res[1] if has_noise else NULL, n_pol=input.n_pol)
    output.execution_time = time.time() - t0
    return back(output)


# ----------------------------------------------------------------------------- receiver front-end
# SURVEY.md 8(f) rank 2.  The random currents / fields are drawn on the host from the GLOBAL np.random
# generator with the reference's calls in the reference's order, so a script that seeds np.random gets the
# same realisation from either library, and uploaded; all arithmetic (square law, gain, the sums, the zero-phase
# filters) runs on the GPU and the results stay there.
# rng="device": the library's Philox4x32-10 generator (csrc/device_mem.hip) instead of NumPy's global one -- no
# seed-for-seed parity with the reference then, only its statistics, but nothing leaves the GPU.  Every draw
# takes the next stream of the current seed; `device_rng_seed` restarts the sequence (default: OS entropy).
_DEVICE_RNG = {"seed": int.from_bytes(os.urandom(8), "little"), "stream": 0}


def device_rng_seed(seed: int):
    """Seed the device generator used by ``PD`` / ``EDFA`` / ``LASER`` with ``rng="device"``."""
    _DEVICE_RNG["seed"], _DEVICE_RNG["stream"] = int(seed), 0


def _device_randn(shape, std, dtype, dev):
    _DEVICE_RNG["stream"] += 1
    return _lib.randn_device(shape, std, _DEVICE_RNG["seed"], _DEVICE_RNG["stream"], dtype, dev)


def _check_rng(rng):
    if rng not in ("numpy", "device"):
        raise ValueError("`rng` must be 'numpy' (the reference's draws, on the host) or 'device'")


_PD_MODES = ("ase-only", "thermal-only", "shot-only", "ase-thermal", "ase-shot", "thermal-shot", "all", "none")


def _idb(x):
    """dB -> linear (reference ``utils.py:422-451``)."""
    return 10 ** (x / 10)


def PD(input: optical_signal, BW: float, r: float = 1.0, T: float = 300.0, R_load: float = 50.0,
       include_noise: str = "all", i_dark: float = 10e-9, Fn=0, *, device=None, rng: str = "numpy") -> electrical_signal:
    """P-I-N photodetector (reference ``devices.py:1378-1555``): ``v = R_load * (r * |E|^2 + noise)``
    summed over the polarisations, low-pass filtered to ``BW`` [Hz].

    The input's own noise (ASE) appears as signal-ASE and ASE-ASE beat currents in ``.noise``; thermal
    (``4 kB T Fn fs/2 / R_load``) and shot (``2 e (mean(i_ph) + i_dark) fs/2``) currents are Gaussian.
    ``include_noise`` selects the terms as in the reference ('all', 'none', 'ase-only', 'thermal-only',
    'shot-only', 'ase-thermal', 'ase-shot', 'thermal-shot'; case-insensitive).

    ``rng`` (extension): ``"numpy"`` draws the Gaussian currents from NumPy's global generator with the
    reference's calls in the reference's order (seed-for-seed identical output: the default, because it is the
    reference's result sample for sample); ``"device"`` uses the library's Philox generator on the GPU -- same
    statistics, the whole detector stays on the device: 1.7 ms instead of 22 ms for ``PD('all')`` at 2^20 x 2
    (15x; measured, profiles/r04_final_cfg_times.txt).  Use it wherever the realisation need not be the reference's.
    """
    from numbers import Real
    from scipy.constants import e, k as kB
    t0 = time.time()
    input, grid, back = _adopt(input, "optical_signal")
    if not isinstance(input, optical_signal):
        raise TypeError("`input` must be of type 'optical_signal'.")
    if not isinstance(r, Real):
        raise TypeError("`r` must be a scalar value.")
    if r <= 0 or r > 1:
        raise ValueError("`r` must be in the range (0,1]")
    if not isinstance(T, Real):
        raise TypeError("`T` must be a scalar value.")
    if T < 0:
        raise ValueError("`T` must be a positive value.")
    if not isinstance(R_load, Real):
        raise TypeError("`R_load` must be a scalar value.")
    if R_load < 0:
        raise ValueError("`R_load` must be a positive value.")
    if not isinstance(include_noise, str):
        raise TypeError("`include_noise` must be a string.")
    mode = include_noise.lower()
    _check_rng(rng)
    if mode not in _PD_MODES:
        raise ValueError("The argument `include_noise` must be one of the following: 'ase-only','thermal-only','shot-only',"
                         "'ase-thermal','ase-shot','thermal-shot','all', 'none'.")
    dev = default_device() if device is None else int(device)
    raw_s, raw_n = input._raw("signal"), input._raw("noise")
    has_ase = raw_n is not NULL
    sig_d = _dev_array(raw_s, np.complex128, dev)
    if mode == "none":
        # nothing random to add: detector and filter back to back
        v, _ = _lib.square_law_device(sig_d, None, r, post=R_load)
        output = LPF(electrical_signal.from_device(v), BW, fs=grid.fs, device=dev)
        output.execution_time = time.time() - t0
        return back(output)
    # photocurrents r |E|^2 (signal) and the signal-ASE / ASE-ASE beat terms, summed over the polarisations
    ds, dn = _lib.square_law_device(sig_d, _dev_array(raw_n, np.complex128, dev) if has_ase else None, r)

    def normal(std):
        """Gaussian current of the given standard deviation: the device generator, or NumPy's global one with the
        reference's call (devices.py:1521-1527) uploaded as it is."""
        if rng == "device":
            return _device_randn(ds.shape, std, np.float64, dev)
        return _lib.DeviceArray.from_host(np.random.normal(0, std, input.size), np.float64, dev)
    d_T = d_N = None
    if "thermal" in mode or mode == "all":
        d_T = normal((4 * kB * T * grid.fs / 2 * _idb(Fn) / R_load) ** 0.5)
    if "shot" in mode or mode == "all":
        mean = _lib.mean_device(ds, dn if has_ase else None)
        d_N = normal((2 * e * (mean + i_dark) * grid.fs / 2) ** 0.5)
    d_ase = dn if (has_ase and ("ase" in mode or mode == "all")) else None
    # (ase + shot) + thermal + dark current, times the load: the reference's order of additions (devices.py:1530-1549)
    v_noise = _lib.sum3_device(d_ase, d_N, d_T, i_dark, R_load, ds)       # all three absent ('ase-only', noiseless input): dark current alone
    v_sig = _lib.sum3_device(ds, None, None, 0.0, R_load, ds)
    output = LPF(electrical_signal.from_device(v_sig, v_noise), BW, fs=grid.fs, device=dev)
    output.execution_time = time.time() - t0
    return back(output)


def EDFA(input: optical_signal, G: float, NF: float, BW: float = None, *, device=None, rng: str = "numpy") -> optical_signal:
    """Erbium-doped fibre amplifier, simplest model (reference ``devices.py:829-942``): field gain
    ``sqrt(G)``, ASE of power ``NF h f0 (G - 1) fs`` split over two polarisations x two quadratures
    added to ``.noise``, then an optical ``BPF`` of bandwidth ``BW`` if given.  Output is always
    dual-polarisation (a single-polarisation input gets an empty y signal, but ASE in both).

    ``rng`` as in :func:`PD`: ``"numpy"`` (default) draws the ASE field from NumPy's global generator exactly as the
    reference does (seed-for-seed identical), ``"device"`` from the library's Philox generator on the GPU: 1.5 ms
    instead of 54 ms at 2^20 x 2 (35x; profiles/r04_final_cfg_times.txt) -- the four 2^20-sample host draws are
    all of the cost."""
    from scipy.constants import h
    t0 = time.time()
    input, grid, back = _adopt(input, "optical_signal")
    if not isinstance(input, optical_signal):
        raise TypeError("`input` must be of type 'optical_signal'.")
    _check_rng(rng)
    g = np.sqrt(_idb(G))
    # gain, ASE loading and the optical filter on the GPU; only the source of the Gaussian field differs with `rng`
    dev = default_device() if device is None else int(device)
    raw_s, raw_n = input._raw("signal"), input._raw("noise")
    n = input.size

    def two_rows(a):                                      # (N,) -> (2, N) with an empty y polarisation; (2, N) as it is
        d = _dev_array(a, np.complex128, dev)
        if d.ndim == 2:
            return _lib.scale_add_device(d, g)
        out2 = _lib.zeros_device((2, n), np.complex128, dev)                      # empty y polarisation
        x = _lib.scale_add_device(d, g)
        _lib._check(_lib.load().ssfm_device_copy(dev, _lib._VP(out2.ptr), _lib._VP(x.ptr), n * 16, 2), "ssfm_device_copy")
        return out2
    sig = two_rows(raw_s)
    P_ase = _idb(NF) * h * grid.f0 * (_idb(G) - 1) * grid.fs
    if rng == "device":
        ase = _device_randn((2, n), np.sqrt(P_ase / 4), np.complex128, dev)
    else:                                                 # the reference's draws (devices.py:930-931), uploaded as they are
        draws = np.sqrt(P_ase / 4) * np.random.randn(4, n)
        ase = _lib.DeviceArray.from_host(draws[:2] + 1j * draws[2:], np.complex128, dev)
    noi = ase if raw_n is NULL else two_rows(raw_n) + ase
    output = _wrap_out(optical_signal, sig, noi, n_pol=2)
    if BW is not None:
        output = _bpf_on_grid(output, BW, grid, device)
    output.execution_time = time.time() - t0
    return back(output)


# ----------------------------------------------------------------------------- PRBS
_PRBS_TAPS = {7: (7, 6), 9: (9, 5), 11: (11, 9), 15: (15, 14), 20: (20, 3), 23: (23, 18), 31: (31, 28)}


def PRBS(order: int, len: int = None, seed: int = None, return_seed: bool = False, *, device=None):
    """Pseudo-random binary sequence of the reference's LFSR (``devices.py:63-182``), bit for bit: polynomial
    ``x^order + x^t2 + 1`` (PRBS-7/9/11/15/20/23/31), state seeded with ``seed % 2**order`` (default all ones;
    0 becomes 1 with a UserWarning), output = bit 0 of the state before every shift.  With ``return_seed`` also
    the final state, to continue the sequence.

    The reference walks the register one bit per Python iteration.  Here the bits are generated by a HIP kernel
    (``ssfm_prbs``, csrc/prbs.hip): a shift is a linear map over GF(2), so every thread jumps to its chunk of the
    sequence with powers of the shift matrix and walks it exactly as the reference does.  The sequence stays in GPU
    memory (``binary_sequence.from_device``) until somebody reads ``.data``; a ``DAC`` takes it from there.
    """
    import warnings
    t0 = time.time()
    if seed is not None:
        seed = seed % (2 ** order)
    else:
        seed = (1 << order) - 1
    if seed == 0:
        seed = 1
        warnings.warn("The seed can't be 0 or a multiple of 2**order. It has been changed to 1.", UserWarning)
    if len is not None:
        if not isinstance(len, int):
            raise TypeError("The parameter `len` must be an integer.")
        if len <= 0:
            raise ValueError("The parameter `len` must be an integer greater than cero.")
    else:
        len = 2 ** order - 1
    if order not in _PRBS_TAPS:
        raise ValueError("The parameter `order` must be one of the following values (7, 9, 11, 15, 20, 23, 31).")
    dev = default_device() if device is None else int(device)
    bits, last = _lib.prbs_device(order, len, seed, dev)
    output = binary_sequence.from_device(bits)
    if not KEEP_ON_DEVICE:
        output.data                                         # noqa: B018 -- the attribute read downloads
    output.execution_time = time.time() - t0
    if not return_seed:
        return output
    return output, last


# ----------------------------------------------------------------------------- DAC
_DAC_SHAPES = ["nrz", "gaussian", "rcos"]


def _pulse_grid(span, sps):
    """``t = linspace(-span/2, span/2, span*sps + 1)`` as (npts, start, step, stop), NumPy's own arithmetic."""
    npts = span * sps + 1
    start, stop = -span / 2, span / 2
    return npts, start, (stop - start) / (npts - 1), stop


def _nrz_spec(span, sps, T):
    """reference ``utils.py:1924-1947``: 1 on [-T/2, T/2)."""
    return (0,) + _pulse_grid(span, sps) + (0, [-T / 2, T / 2]), False


def _gauss_spec(span, sps, T=1, m=1, c=0.0):
    """reference ``utils.py:1880-1922``: ``exp(-(alpha (1 + j c) t)^(2m))``, ``alpha = 2 sqrt(ln 2) / T``; complex."""
    if 2 * m > 98:
        raise ValueError("The parameter `m` must be below 50.")
    k = 2 * np.sqrt(np.log(2)) / T * (1 + 1j * c)
    return (1,) + _pulse_grid(span, sps) + (2 * m, [k.real, k.imag]), True


def _rcos_spec(beta, span, sps, shape="sqrt"):
    """reference ``utils.py:1791-1878`` (MATLAB ``rcosdesign`` without the energy normalisation)"""
    if not (0 <= beta <= 1):
        raise ValueError("beta must be in [0, 1]")
    if shape not in ("sqrt", "normal"):
        raise ValueError("shape must be 'sqrt' or 'normal'")
    grid = _pulse_grid(span, sps)
    if beta == 0:
        return (4,) + grid + (0, []), False
    if shape == "normal":
        return (2,) + grid + (0, [2 * beta, np.pi * beta, (np.pi / 4) * np.sinc(1 / (2 * beta))]), False
    at_zero = (1 - beta) + 4 * beta / np.pi
    at_special = (beta / np.sqrt(2)) * ((1 + 2 / np.pi) * np.sin(np.pi / (4 * beta)) + (1 - 2 / np.pi) * np.cos(np.pi / (4 * beta)))
    return (3,) + grid + (0, [beta, 4 * beta, 1 - beta, 1 + beta, 1 / (4 * beta), at_zero, at_special]), False


def _upfir_device(bits, h, up: int, dev: int, spec=None) -> "_lib.DeviceArray":
    """``upfir`` of the reference (``utils.py:1949-1981``): zero-stuffing at offset ``up//2`` and the 'same' part of
    the linear convolution with the pulse -- as ONE circular convolution on a power-of-two complex128 plan
    (``x <- ifft(fft(x) * fft(h))``, three launches); SciPy's ``fftconvolve`` does the same on the host.
    The pulse is either an explicit impulse response ``h`` (uploaded) or ``spec = (args of Plan.load_pulse, complex?)``
    of a built-in shape, generated on the device.  ``bits``: a host array, or the uint8 device array a ``PRBS`` left
    in GPU memory (then nothing is uploaded at all)."""
    n = int(bits.size) * up
    if spec is None:
        h = np.asarray(h)
        taps, cplx = h.size, np.iscomplexobj(h)
    else:
        taps, cplx = spec[0][1], spec[1]
    full = n + taps - 1
    M = 1 << max(8, (full - 1).bit_length())
    lo, hi = _lib.supported_log2n(_lib.C128, direct=True)
    if M > (1 << hi):
        raise ValueError(f"DAC: {bits.size} bits x {up} samples with a {taps}-tap pulse exceed the device path (2^{hi} points)")
    plan = get_plan(M, 1, _lib.C128, dev)
    with plan.lock:
        if spec is None:
            hd = _lib.DeviceArray.from_host(np.ascontiguousarray(h, dtype=np.complex128 if cplx else np.float64), None, dev)
            plan.load_padded(hd)                                # field <- h, zero-padded
        else:
            plan.load_pulse(*spec[0])                           # field <- the pulse, generated in place
        plan.table_from_field(0)                                # slot 0 <- fft(h)  (the plan drops the slot's old label itself)
        if _on_device(bits) and bits.dtype == np.uint8:
            plan.load_bits(bits, up)                            # field <- zero-stuffed bits, where the generator left them
        else:
            bd = _lib.DeviceArray.from_host(np.ascontiguousarray(bits, dtype=np.float64), np.float64, dev)
            plan.load_symbols(bd, up)                           # field <- zero-stuffed bits
        plan.apply_table(0)
        out = _lib.DeviceArray((n,), np.complex128, dev)
        plan.copy_from_field(((taps - 1) // 2) * 16, out.ptr, n * 16)          # 'same': centred with respect to the full output
    return out if cplx else _lib.real_device(out)


def DAC(input, pulse_shape: str = "nrz", coupling: str = "DC", Vpp: float = 1.0, offset: float = 0.0, h=None, BW: float = None,
        *, device=None, **kwargs) -> electrical_signal:
    """Digital-to-analog converter (reference ``devices.py:185-350``): a bit sequence becomes ``gv.sps`` samples
    per bit of the chosen pulse (``'nrz'`` with ``T``; ``'gaussian'`` with ``T``, ``m``, ``c``; ``'rcos'`` with
    ``beta``, ``rcos_type``; or an explicit impulse response ``h``), scaled by ``Vpp``, shifted by ``offset``,
    optionally AC-coupled and band-limited by ``LPF(BW)``.  The pulse spans ``max(4, bits - 4)`` symbols, i.e. the
    shaping is a convolution as long as the signal: it runs as an FFT convolution on the GPU."""
    t0 = time.time()
    seq = input if isinstance(input, binary_sequence) else binary_sequence(input)
    bits = seq.size
    sps = gv.sps
    dev = default_device() if device is None else int(device)
    data = seq._raw()                                       # host bits, or the device array a PRBS left in GPU memory
    if _on_device(data) and data.device != dev:
        data = seq.to_numpy()
    span = max(4, bits - 4)
    spec = None
    if h is not None:
        h = np.asarray(h)
    elif pulse_shape.lower() not in _DAC_SHAPES:
        raise ValueError(f"The parameter `pulse_shape` must be one of the following values {_DAC_SHAPES}")
    elif pulse_shape.lower() == "nrz":
        T = kwargs.get("T", 1)
        if not isinstance(T, int):
            raise TypeError("The parameter `T` must be an integer.")
        if T <= 0:
            raise ValueError("The parameter `T` must be greater than 0.")
        if T > 2 * sps:
            raise ValueError("The parameter `T` must be less than 2*sps.")
        spec = _nrz_spec(span, sps, T)
    elif pulse_shape.lower() == "gaussian":
        c, m, T = kwargs.get("c", 0.0), kwargs.get("m", 1), kwargs.get("T", 1)
        if not isinstance(c, (int, float)):
            raise TypeError("The parameter `c` must be a real number.")
        if not isinstance(m, int):
            raise TypeError("The parameter `m` must be an integer.")
        if not isinstance(T, int):
            raise TypeError("The parameter `T` must be an integer.")
        if m <= 0:
            raise ValueError("The parameter `m` must be greater than 0.")
        if T <= 0:
            raise ValueError("The parameter `T` must be greater than 0.")
        if T > 2 * sps:
            raise ValueError("The parameter `T` must be less than 2*sps.")
        spec = _gauss_spec(span, sps, T=T, m=m, c=c)
    else:
        spec = _rcos_spec(kwargs.get("beta", 0.25), span, sps, shape=kwargs.get("rcos_type", "normal"))
    if Vpp is not None:
        if not isinstance(Vpp, (int, float)):
            raise TypeError("The parameter `Vpp` must be a scalar value.")
        if Vpp <= 0 or Vpp > 48:
            raise ValueError("The parameter `Vpp` must be in the range (0, 48] Volts.")
    if offset is not None:
        if not isinstance(offset, (int, float)):
            raise TypeError("The parameter `offset` must be a scalar value.")
        if np.abs(offset) > 48:
            raise ValueError("The parameter `offset` must be in the range [-48, 48] Volts.")
    if coupling.upper() not in ("AC", "DC"):
        raise ValueError("The parameter `coupling` must be either 'AC' or 'DC'.")
    x = _upfir_device(data, h, sps, dev, spec)              # stays on the device: float64, or complex128 for a complex pulse
    if Vpp is not None:
        x = _lib.axpb_device(x, Vpp, 0.0)
    if offset is not None:
        x = _lib.axpb_device(x, 1.0, offset)
    if coupling.upper() == "AC":
        x = _lib.shift_device(x, -_lib.mean2_device(x))       # x - mean(x), real or complex, on the device
    output = _wrap_out(electrical_signal, x, NULL)
    if BW is not None:
        output = LPF(output, BW, device=dev)
    output.execution_time = time.time() - t0
    return output


# ----------------------------------------------------------------------------- LASER / MZM
# Elementwise HIP kernels (csrc/transmitter.hip) in the reference's own order of operations; the laser's random
# increments are drawn from NumPy's global generator with the reference's calls, so a seeded script gets the same laser
# noise.  They complete the transmitter chain PRBS -> DAC -> MZM(LASER) in front of FIBER, which stays in GPU memory.
def LASER(P0: float, lw: float = None, rin: float = None, df: float = None, *, device=None, rng: str = "numpy") -> optical_signal:
    """CW laser of ``P0`` dBm over ``gv.t`` (reference ``devices.py:353-510``): optional linewidth ``lw`` [Hz] (Wiener
    phase noise), relative intensity noise ``rin`` [dB/Hz] and frequency offset ``df`` [Hz]; single polarisation.
    Real-valued (float64) unless ``lw`` or ``df`` is given, as in the reference.  ``rng`` as in :func:`PD`:
    ``"numpy"`` draws the noise from NumPy's global generator with the reference's calls, ``"device"`` from the
    library's Philox generator (the running sum of the phase increments is then a device scan too)."""
    t0 = time.time()
    _check_rng(rng)
    t = gv.t
    n = t.size
    dev = default_device() if device is None else int(device)
    amp = float(np.sqrt(10 ** (P0 / 10 - 3)))
    phase = rin_noise = w = None
    if lw is not None:
        std = np.sqrt(2 * np.pi * lw * gv.dt)
        if rng == "device":
            phase = _lib.cumsum_device(_device_randn((n,), std, np.float64, dev))
        else:
            phase = _lib.DeviceArray.from_host(np.cumsum(np.random.normal(0, std, n)), np.float64, dev)
    if rin is not None:
        std = np.sqrt(_idb(rin) * gv.fs)
        if rng == "device":
            rin_noise = _device_randn((n,), std, np.float64, dev)
            lowest = _lib.min_device(rin_noise)
        else:
            r = np.random.normal(0, std, n)
            lowest = r.min()
            rin_noise = _lib.DeviceArray.from_host(r, np.float64, dev) if lowest >= -1 else None
        if lowest < -1:
            raise ValueError("Noise power is to high, try decrease RIN parameter.")
    if df is not None:
        if np.abs(df) > gv.fs / 2:
            raise ValueError("The laser frequency is out of the Nyquist range. Try increase the sampling frequency.")
        w = 2 * np.pi * df                                   # exp(1j*2*pi*df*t): the phase is (2 pi df) * t_i
    out = _lib.laser_device(n, amp, phase, rin_noise, w, float(t[1]) if n > 1 else 0.0, float(t[-1]), dev)
    output = _wrap_out(optical_signal, out, NULL)
    output.execution_time = time.time() - t0
    return output


def MZM(op_input: optical_signal, el_input, bias: float = 0.0, Vpi: float = 5.0, loss_dB: float = 0.0, ER_dB: float = 26.0,
        pol: str = "x", BW: float = None, *, device=None) -> optical_signal:
    """Mach-Zehnder modulator (reference ``devices.py:620-786``): ``out = in * sqrt(loss) (cos g + j eta/2 sin g)``,
    ``g = pi/(2 Vpi) (v + bias)``, ``eta = 2 sqrt(1/ER)``; the drive voltage's own noise enters ``g``, optical
    signal and optical noise are both multiplied; of a dual-polarisation input only ``pol`` survives; ``BW`` adds
    a ``BPF``."""
    t0 = time.time()
    op_input, grid, back = _adopt(op_input, "optical_signal")
    if not isinstance(op_input, optical_signal):
        raise TypeError("`op_input` must be of type 'optical_signal'.")
    if isinstance(el_input, electrical_signal) or type(el_input).__name__ == "electrical_signal":
        el_input = _adopt(el_input, "electrical_signal")[0]
    else:
        el_input = electrical_signal(el_input)
    if el_input.ndim > 1:
        raise ValueError("`el_input` must be a scalar or 1D-array.")
    if pol not in ["x", "y"]:
        raise ValueError("The parameter `pol` must be one of the following values ('x', 'y').")
    loss = _idb(-loss_dB)
    eta = 2 * _idb(-ER_dB) ** 0.5
    k = np.pi / 2 / Vpi
    raw_s, raw_n = op_input._raw("signal"), op_input._raw("noise")
    raw_v, raw_vn = el_input._raw("signal"), el_input._raw("noise")
    dev = default_device() if device is None else int(device)
    n = tuple(raw_s.shape)[-1]
    # one HIP kernel (ssfm_mzm, csrc/transmitter.hip) whatever the inputs are: a DAC output or a carrier that is
    # already in GPU memory is used where it lies, host arrays are uploaded (a scalar drive is spread over the grid)
    cplx = any(a is not NULL and a.dtype.kind == "c" for a in (raw_v, raw_vn))
    vdt = np.complex128 if cplx else np.float64

    def drive(a):
        if a is NULL:
            return None
        if _on_device(a) and tuple(a.shape) == (n,) and a.dtype == vdt:
            return _dev_array(a, vdt, dev)
        host = a.to_host() if _on_device(a) else np.asarray(a)
        return _lib.DeviceArray.from_host(np.broadcast_to(host, (n,)), vdt, dev)     # ValueError when the lengths disagree
    out_s, out_n = _lib.mzm_device(_dev_array(raw_s, np.complex128, dev), None if raw_n is NULL else _dev_array(raw_n, np.complex128, dev),
                                   drive(raw_v), drive(raw_vn), k, bias, loss ** 0.5, eta / 2, 1 if pol == "x" else 0)
    output = _wrap_out(optical_signal, out_s, NULL if out_n is None else out_n, n_pol=op_input.n_pol)
    if BW is not None:
        output = _bpf_on_grid(output, BW, grid, device)
    output.execution_time = time.time() - t0
    return back(output)
