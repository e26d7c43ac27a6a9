"""ctypes binding of the C ABI in ``include/ssfm_amd.h`` (built as ``_ssfm_amd.so``).

There is no CPU fallback: if the shared library is missing, or no gfx950 device is
visible, every entry point raises :class:`SsfmError` -- loudly, by design.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SSFM_LIB") or os.path.join(_HERE, "_ssfm_amd.so")   # SSFM_LIB: dev override

C64, C128 = 0, 1
F64_REAL = 2                           # ssfm_device_convert only: real float64 values
HOST_PINNED = -1                       # ssfm_device_alloc / _free: a page-locked host buffer
REDUCE_MEAN, REDUCE_MEAN2, REDUCE_POWER, REDUCE_MIN = 0, 1, 2, 3      # ssfm_device_reduce
_CDTYPE = {C64: np.complex64, C128: np.complex128}
_RDTYPE = {C64: np.float32, C128: np.float64}

# every symbol include/ssfm_amd.h declares: name -> (restype, argtypes)
_VP, _I, _I64, _D = C.c_void_p, C.c_int, C.c_int64, C.c_double
SYMBOLS = {
    "ssfm_abi_version": (_I, []),
    "ssfm_device_count": (_I, [C.POINTER(_I)]),
    "ssfm_last_error": (C.c_char_p, []),
    "ssfm_supported_log2n": (_I, [_I, C.POINTER(_I), C.POINTER(_I)]),
    "ssfm_plan_create": (_I, [C.POINTER(_VP), _I, _I64, _I, _I]),
    "ssfm_plan_destroy": (_I, [_VP]),
    "ssfm_set_linear_operator": (_I, [_VP, _VP]),
    "ssfm_set_field": (_I, [_VP, _VP, _I]),
    "ssfm_get_field": (_I, [_VP, _VP, _I]),
    "ssfm_field_device_ptr": (_VP, [_VP]),
    "ssfm_propagate_fixed": (_I, [_VP, _D, _VP, _I64, _VP]),
    "ssfm_propagate_adaptive": (_I, [_VP, _D, _D, _D, _I, _I64, C.POINTER(_I64), C.POINTER(_D), _VP]),
    "ssfm_propagate_fixed_capture": (_I, [_VP, _D, _VP, _I64, _VP]),
    "ssfm_adaptive_begin": (_I, [_VP, _D, _D, _D, _I, _I64, _I]),
    "ssfm_adaptive_run": (_I, [_VP, _I64, _VP, C.POINTER(_I64), C.POINTER(_I)]),
    "ssfm_adaptive_finish": (_I, [_VP, C.POINTER(_I64), C.POINTER(_D)]),
    "ssfm_adaptive_set_capture": (_I, [_VP, _VP]),
    "ssfm_apply_transfer": (_I, [_VP, _VP]),
    "ssfm_apply_dispersion": (_I, [_VP, _D, _D, _VP]),
    "ssfm_sosfiltfilt": (_I, [_I, _VP, _VP, _I, _VP, _VP, _I64, _I, _I, _I]),
    "ssfm_sosfiltfilt_last": (_I, [_I, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "ssfm_square_law": (_I, [_I, _VP, _VP, _I, _I64, _D, _D, _VP, _VP, _I]),
    "ssfm_device_alloc": (_I, [_I, C.c_size_t, C.POINTER(_VP)]),
    "ssfm_device_free": (_I, [_I, _VP, C.c_size_t]),
    "ssfm_device_copy": (_I, [_I, _VP, _VP, C.c_size_t, _I]),
    "ssfm_device_convert": (_I, [_I, _VP, _I, _VP, _I, _I64]),
    "ssfm_device_add": (_I, [_I, _VP, _VP, _VP, _I, _I64]),
    "ssfm_device_randn": (_I, [_I, _VP, _I64, C.c_uint64, C.c_uint64, _D, _D]),
    "ssfm_device_sum3": (_I, [_I, _VP, _VP, _VP, _VP, _D, _D, _I64]),
    "ssfm_device_scale_add": (_I, [_I, _VP, _VP, _D, _VP, _I64]),
    "ssfm_device_cumsum": (_I, [_I, _VP, _VP, _I64]),
    "ssfm_mzm": (_I, [_I, _VP, _VP, _VP, _VP, _I, _I64, _VP, _VP, _I, _D, _D, _D, _D, _I]),
    "ssfm_device_axpb": (_I, [_I, _VP, _VP, _D, _D, _I64, _I]),
    "ssfm_device_mem_info": (_I, [_I, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "ssfm_transfer_table": (_I, [_VP, _VP, _I]),
    "ssfm_apply_table": (_I, [_VP, _I]),
    "ssfm_load_padded": (_I, [_VP, _I64, _VP, _I, _I64]),
    "ssfm_load_symbols": (_I, [_VP, _I64, _VP, _I, _I64, _I]),
    "ssfm_laser": (_I, [_I, _VP, _I64, _D, _VP, _VP, _I, _D, _D, _D]),
    "ssfm_load_pulse": (_I, [_VP, _I64, _I, _I64, _D, _D, _D, _I, _VP]),
    "ssfm_table_from_field": (_I, [_VP, _I]),
    "ssfm_chirp_propagate": (_I, [_VP, _I64, _I, _VP, _VP, _VP, _VP, _I64, _D, C.POINTER(_D), _I64, _D, _D, _I, _I64, C.POINTER(_D), C.POINTER(_I64)]),
    "ssfm_synchronize": (_I, [_VP]),
    "ssfm_stream": (_VP, [_VP]),
    "ssfm_last_propagate_ms": (_I, [_VP, C.POINTER(C.c_float), C.POINTER(_I64)]),
    "ssfm_last_run_info": (_I, [_VP, _VP, C.c_size_t]),
    "ssfm_set_profiling": (_I, [_VP, _I]),
    "ssfm_kernel_times": (_I, [_VP, C.POINTER(_I64), C.POINTER(_D)]),
    "ssfm_plan_set_tag": (_I, [_VP, _I, C.c_uint64]),
    "ssfm_plan_get_tag": (_I, [_VP, _I, C.POINTER(C.c_uint64)]),
    "ssfm_device_chirp": (_I, [_I, _VP, _I64, _I]),
    "ssfm_chirp_setup": (_I, [_VP, _I64, _I64]),
    "ssfm_chirp_propagate_c64": (_I, [_VP, _VP, _VP, _VP, _I64, _D, C.POINTER(_D), _I64, _D, _D, _I64, C.POINTER(_D), C.POINTER(_I64)]),
    "ssfm_chirp_transfer": (_I, [_VP, _I64, _I, _VP, _VP, _VP, _I64, _I]),
    "ssfm_chirp_fourier": (_I, [_VP, _I64, _I, _VP, _VP, _VP, _I64, _I]),
    "ssfm_device_reduce": (_I, [_I, _I, _VP, _VP, _I, _I64, _I, C.POINTER(_D)]),
    "ssfm_debug": (_I, [_VP, _I, _I64, _VP]),
    "ssfm_device_shift": (_I, [_I, _VP, _VP, _I64, _I, _D, _D]),
    "ssfm_prbs": (_I, [_I, _VP, _I64, _I, C.c_uint32, C.POINTER(C.c_uint32)]),
    "ssfm_load_qpsk": (_I, [_VP, _I64, _I, _VP, _I64, _I]),
}


# enum ssfm_engine of include/ssfm_amd.h, by value
ENGINES = ("none", "two_kernel", "small", "medium", "adaptive_3_launches", "adaptive_fused", "small_adaptive", "medium_adaptive",
           "chirp_small", "chirp_small_adaptive", "chirp_steps", "chirp_medium", "chirp_medium_adaptive", "split", "split_adaptive")
DIRECT_LOG2_MAX = 22      # rows of more than 2^22 samples are split plans (csrc/ssfm_split.hpp): propagation, DM and field transfers -- no tables, chirp-z or strided capture


class AdaptiveCapture(C.Structure):
    """``ssfm_adaptive_capture`` of include/ssfm_amd.h."""
    _fields_ = [("every", C.c_int64), ("steps", C.POINTER(C.c_int64)), ("n_steps", C.c_int64), ("fields", C.c_void_p), ("capacity", C.c_int64),
                ("taken", C.POINTER(C.c_int64)), ("n_taken", C.POINTER(C.c_int64))]


class Capture(C.Structure):
    """``ssfm_capture`` of include/ssfm_amd.h."""
    _fields_ = [("every", C.c_int64), ("fields", C.c_void_p), ("scalars", C.c_void_p)]


class RunInfo(C.Structure):
    """``ssfm_run_info`` of include/ssfm_amd.h."""
    _fields_ = [("engine", C.c_int), ("fell_back", C.c_int), ("fallbacks_total", C.c_int64), ("lanes", C.c_int), ("lanes_configured", C.c_int), ("lanes_share_queue", C.c_int),
                ("lanes_remade", C.c_int), ("lanes_dropped", C.c_int), ("lane_heals", C.c_int), ("lane_alone_us", C.c_float), ("lane_pair_us", C.c_float),
                ("lane_last_us", C.c_float), ("lane_score", C.c_float), ("lanes_from_pool", C.c_int), ("pad_", C.c_int), ("lane_ratings_total", C.c_int64),
                ("lane_pairs_reused", C.c_int64)]


class SsfmError(RuntimeError):
    """A call into the HIP library failed (or the library itself is missing)."""


_lib = None


def load():
    """Load ``_ssfm_amd.so`` and bind every declared symbol.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SsfmError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C opticomlib_amd/csrc`.  There is no CPU fallback for the HIP path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.ssfm_abi_version() != 3:
        raise SsfmError(f"ABI version mismatch: library reports {lib.ssfm_abi_version()}, binding expects 3")
    _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        msg = load().ssfm_last_error().decode(errors="replace")
        raise SsfmError(f"{what} failed (status {rc}): {msg}")


def _check_filter(rc, what):
    """An input shorter than the padding is the caller's ValueError in SciPy (``sosfiltfilt`` / ``_validate_pad``), and so
    it is here; everything else is a library failure."""
    if rc == 1:
        msg = load().ssfm_last_error().decode(errors="replace")
        if "padlen" in msg:
            raise ValueError(msg)
    _check(rc, what)


def device_count() -> int:
    n = _I(0)
    rc = load().ssfm_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def supported_log2n(precision=C64, direct=False):
    """(lo, hi) of log2(samples per row) a plan takes; ``direct``: of the plans whose field buffer is one line (what the chirp-z path and the transfer tables need)."""
    lo, hi = _I(0), _I(0)
    _check(load().ssfm_supported_log2n(precision, C.byref(lo), C.byref(hi)), "ssfm_supported_log2n")
    return lo.value, (min(hi.value, DIRECT_LOG2_MAX) if direct else hi.value)


def _ptr(a: np.ndarray):
    return C.c_void_p(a.ctypes.data)


def sosfiltfilt(sos: np.ndarray, zi: np.ndarray, x: np.ndarray, device: int = 0) -> np.ndarray:
    """Zero-phase SOS filtering of the last axis of ``x`` (float64 or complex128) on the GPU."""
    sos = np.ascontiguousarray(sos, dtype=np.float64)
    zi = np.ascontiguousarray(zi, dtype=np.float64)
    x = np.asarray(x)
    is_c = np.iscomplexobj(x)
    xs = np.ascontiguousarray(x, dtype=np.complex128 if is_c else np.float64)
    n = xs.shape[-1]
    batch = int(xs.size // n) if n else 0
    y = np.empty_like(xs)
    _check_filter(load().ssfm_sosfiltfilt(int(device), _ptr(sos), _ptr(zi), sos.shape[0], _ptr(xs), _ptr(y), n, batch, int(is_c), 0),
                  "ssfm_sosfiltfilt")
    return y


def sosfiltfilt_device(sos: np.ndarray, zi: np.ndarray, x_ptr: int, y_ptr: int, n: int, batch: int, is_complex: bool,
                       device: int = 0) -> None:
    """The same on DEVICE buffers (raw pointers; float64 or interleaved complex128, ``batch`` rows of ``n``)."""
    sos = np.ascontiguousarray(sos, dtype=np.float64)
    zi = np.ascontiguousarray(zi, dtype=np.float64)
    _check_filter(load().ssfm_sosfiltfilt(int(device), _ptr(sos), _ptr(zi), sos.shape[0], _VP(x_ptr), _VP(y_ptr), int(n), int(batch),
                                          int(bool(is_complex)), 1), "ssfm_sosfiltfilt")


def square_law(signal: np.ndarray, noise, r: float, device: int = 0):
    """``r * (x * x.conj()).real`` summed over the polarisations, signal and noise currents apart
    (``noise`` may be None).  ``signal``: (N,) or (2, N) complex."""
    s = np.ascontiguousarray(signal, dtype=np.complex128)
    n = s.shape[-1]
    n_pol = 1 if s.ndim == 1 else s.shape[0]
    i_sig = np.empty(n, dtype=np.float64)
    if noise is None:
        _check(load().ssfm_square_law(int(device), _ptr(s), None, n_pol, n, float(r), 1.0, _ptr(i_sig), None, 0), "ssfm_square_law")
        return i_sig, None
    nz = np.ascontiguousarray(noise, dtype=np.complex128)
    if nz.shape != s.shape:
        raise ValueError(f"signal and noise shapes differ: {s.shape} vs {nz.shape}")
    i_noise = np.empty(n, dtype=np.float64)
    _check(load().ssfm_square_law(int(device), _ptr(s), _ptr(nz), n_pol, n, float(r), 1.0, _ptr(i_sig), _ptr(i_noise), 0), "ssfm_square_law")
    return i_sig, i_noise


def sosfiltfilt_last_ms(device: int = 0) -> float:
    """Device time [ms] of the kernels of the last filter call on ``device``."""
    ms = C.c_float()
    _check(load().ssfm_sosfiltfilt_last(int(device), C.byref(ms), None), "ssfm_sosfiltfilt_last")
    return float(ms.value)


def sosfiltfilt_last_launches(device: int = 0) -> int:
    """Kernel launches of the last filter call on ``device``: 1 (one-launch form) or 3."""
    k = C.c_int()
    _check(load().ssfm_sosfiltfilt_last(int(device), None, C.byref(k)), "ssfm_sosfiltfilt_last")
    return int(k.value)


# ----------------------------------------------------------------------------- device-resident arrays
TRANSFERS = {"h2d": 0, "d2h": 0}        # counts of DeviceArray host<->device copies (tests check laziness with it)


class _PinnedBlock:
    """Page-locked host memory from the library's pool, exposed through the array interface: ``np.asarray(block)`` is
    an ordinary writable NumPy array that keeps the block alive; the block returns to the pool when the array dies."""

    def __init__(self, shape, dtype):
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = _VP()
        _check(load().ssfm_device_alloc(HOST_PINNED, self.nbytes, C.byref(p)), "ssfm_device_alloc")
        self.ptr = int(p.value)
        self.__array_interface__ = {"data": (self.ptr, False), "shape": tuple(shape), "typestr": np.dtype(dtype).str, "version": 3}

    def __del__(self):
        try:
            if self.ptr and _lib is not None:
                _lib.ssfm_device_free(HOST_PINNED, _VP(self.ptr), self.nbytes)
            self.ptr = 0
        except Exception:
            pass


def host_empty(shape, dtype, limit: int = 256 << 20) -> np.ndarray:
    """Uninitialised array for a device-to-host copy: page-locked when that is available (no page faults, no on-the-fly
    locking of the destination by the runtime), plain ``np.empty`` otherwise, for empty shapes and beyond ``limit`` bytes
    (the every-step captures of many GiB stay pageable)."""
    shape = tuple(int(d) for d in shape)
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    if nbytes == 0 or nbytes > limit:
        return np.empty(shape, dtype=dtype)
    try:
        return np.asarray(_PinnedBlock(shape, dtype))
    except SsfmError:
        return np.empty(shape, dtype=dtype)


class DeviceArray:
    """A C-contiguous array in the HBM of one GPU: what a signal object holds between device calls.

    ``dtype`` is complex64, complex128, float64 or uint8 (bit sequences).  The buffer goes back to the library's
    pool when the object is collected.  ``__cuda_array_interface__`` lets torch / RCCL use the memory where it lies."""

    def __init__(self, shape, dtype, device: int = 0):
        self.shape = tuple(int(d) for d in np.atleast_1d(shape)) if not isinstance(shape, tuple) else tuple(int(d) for d in shape)
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.complex64), np.dtype(np.complex128), np.dtype(np.float64), np.dtype(np.uint8)):
            raise TypeError(f"DeviceArray supports complex64, complex128, float64 and uint8, not {self.dtype}")
        self.device = int(device)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self.ptr = 0
        p = _VP()
        _check(load().ssfm_device_alloc(self.device, self.nbytes, C.byref(p)), "ssfm_device_alloc")
        self.ptr = int(p.value)

    # numpy-like metadata (no transfer)
    @property
    def ndim(self):
        return len(self.shape)

    @property
    def __cuda_array_interface__(self):
        return {"shape": self.shape, "typestr": self.dtype.str, "data": (self.ptr, False), "version": 2, "strides": None}

    @property
    def size(self):
        return int(np.prod(self.shape))

    def __len__(self):
        return self.shape[0]

    def free(self):
        if self.ptr and _lib is not None:
            _lib.ssfm_device_free(self.device, _VP(self.ptr), self.nbytes)
        self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    @classmethod
    def from_host(cls, a: np.ndarray, dtype=None, device: int = 0) -> "DeviceArray":
        a = np.ascontiguousarray(a, dtype=dtype)
        d = cls(a.shape, a.dtype, device)
        _check(load().ssfm_device_copy(d.device, _VP(d.ptr), _ptr(a), d.nbytes, 0), "ssfm_device_copy")
        TRANSFERS["h2d"] += 1
        return d

    def to_host(self) -> np.ndarray:
        out = host_empty(self.shape, self.dtype)
        _check(load().ssfm_device_copy(self.device, _ptr(out), _VP(self.ptr), self.nbytes, 1), "ssfm_device_copy")
        TRANSFERS["d2h"] += 1
        return out

    def copy(self) -> "DeviceArray":
        d = DeviceArray(self.shape, self.dtype, self.device)
        _check(load().ssfm_device_copy(self.device, _VP(d.ptr), _VP(self.ptr), self.nbytes, 2), "ssfm_device_copy")
        return d

    def astype(self, dtype) -> "DeviceArray":
        """complex64 <-> complex128, float64 -> either, on the device (a copy when the type already matches)."""
        dtype = np.dtype(dtype)
        if dtype == self.dtype:
            return self.copy()
        codes = {np.dtype(np.complex64): C64, np.dtype(np.complex128): C128, np.dtype(np.float64): 2}
        if self.dtype not in codes or dtype not in codes or codes[dtype] == 2:
            raise TypeError(f"DeviceArray.astype: {self.dtype} -> {dtype} is not supported")
        d = DeviceArray(self.shape, dtype, self.device)
        _check(load().ssfm_device_convert(self.device, _VP(self.ptr), codes[self.dtype], _VP(d.ptr), codes[dtype], self.size), "ssfm_device_convert")
        return d

    def __add__(self, other: "DeviceArray") -> "DeviceArray":
        if not isinstance(other, DeviceArray) or other.shape != self.shape or other.dtype != self.dtype or other.device != self.device:
            raise TypeError("DeviceArray + DeviceArray needs equal shapes, types and devices")
        codes = {np.dtype(np.complex64): C64, np.dtype(np.complex128): C128}
        d = DeviceArray(self.shape, self.dtype, self.device)
        _check(load().ssfm_device_add(self.device, _VP(d.ptr), _VP(self.ptr), _VP(other.ptr), codes[self.dtype], self.size), "ssfm_device_add")
        return d

    def __repr__(self):
        return f"DeviceArray(shape={self.shape}, dtype={self.dtype}, device={self.device})"


def randn_device(shape, std: float, seed: int, stream: int, dtype=np.float64, device: int = 0) -> DeviceArray:
    """``std * N(0, 1)`` from the library's Philox4x32-10 generator (``ssfm_device_randn``).  For a complex dtype
    real and imaginary parts are independent, each of standard deviation ``std``."""
    out = DeviceArray(shape, dtype, device)
    count = out.size * (2 if out.dtype.kind == "c" else 1)
    _check(load().ssfm_device_randn(out.device, _VP(out.ptr), count, int(seed) & (2 ** 64 - 1), int(stream) & (2 ** 64 - 1), 0.0, float(std)),
           "ssfm_device_randn")
    return out


def sum3_device(a, b, c, offset: float, scale: float, like: DeviceArray) -> DeviceArray:
    """``(a + b + c + offset) * scale`` on float64 device arrays (any of a, b, c may be None)."""
    out = DeviceArray(like.shape, np.float64, like.device)
    p = lambda x: None if x is None else _VP(x.ptr)
    _check(load().ssfm_device_sum3(like.device, _VP(out.ptr), p(a), p(b), p(c), float(offset), float(scale), out.size), "ssfm_device_sum3")
    return out


def scale_add_device(a: DeviceArray, factor: float, b=None) -> DeviceArray:
    """``a * factor (+ b)`` for float64 / complex128 device arrays (real ``factor``)."""
    out = DeviceArray(a.shape, a.dtype, a.device)
    count = a.size * (2 if a.dtype.kind == "c" else 1)
    _check(load().ssfm_device_scale_add(a.device, _VP(out.ptr), _VP(a.ptr), float(factor), None if b is None else _VP(b.ptr), count), "ssfm_device_scale_add")
    return out


def mean_device(a: DeviceArray, b=None) -> float:
    m = _D()
    _check(load().ssfm_device_reduce(a.device, REDUCE_MEAN, _VP(a.ptr), None if b is None else _VP(b.ptr), 1, a.size, 0, C.byref(m)), "ssfm_device_reduce")
    return float(m.value)


def axpb_device(a: DeviceArray, alpha: float, beta: float) -> DeviceArray:
    """``a * alpha + beta`` (float64 or complex128 device array; ``beta`` real)."""
    out = DeviceArray(a.shape, a.dtype, a.device)
    _check(load().ssfm_device_axpb(a.device, _VP(out.ptr), _VP(a.ptr), float(alpha), float(beta), a.size, int(a.dtype.kind == "c")), "ssfm_device_axpb")
    return out


def chirp_device(n: int, conj: bool, device: int = 0) -> DeviceArray:
    """``exp(-i pi m^2 / n)`` (or its conjugate), m < n, complex128, generated on the device."""
    out = DeviceArray((int(n),), np.complex128, device)
    _check(load().ssfm_device_chirp(int(device), _VP(out.ptr), int(n), int(bool(conj))), "ssfm_device_chirp")
    return out


def mean2_device(a: DeviceArray):
    """``numpy.mean`` of a float64 / complex128 device array (a Python float or complex)."""
    m = (_D * 2)()
    cplx = a.dtype.kind == "c"
    _check(load().ssfm_device_reduce(a.device, REDUCE_MEAN2, _VP(a.ptr), None, 1, a.size, int(cplx), m), "ssfm_device_reduce")
    return complex(m[0], m[1]) if cplx else float(m[0])


def shift_device(a: DeviceArray, value) -> DeviceArray:
    """``a + value`` (float64 / complex128 device array; complex ``value`` for a complex array)."""
    out = DeviceArray(a.shape, a.dtype, a.device)
    v = complex(value)
    _check(load().ssfm_device_shift(a.device, _VP(out.ptr), _VP(a.ptr), a.size, int(a.dtype.kind == "c"), v.real, v.imag), "ssfm_device_shift")
    return out


def zeros_device(shape, dtype, device: int = 0) -> DeviceArray:
    out = DeviceArray(shape, dtype, device)
    _check(load().ssfm_device_copy(out.device, _VP(out.ptr), None, out.nbytes, 3), "ssfm_device_copy")
    return out


def power_device(ptr: int, rows: int, n: int, is_complex: bool, device: int = 0) -> np.ndarray:
    """Mean ``|x|^2`` of each of ``rows`` rows of ``n`` values at device address ``ptr``."""
    out = (_D * int(rows))()
    _check(load().ssfm_device_reduce(int(device), REDUCE_POWER, _VP(ptr), None, int(rows), int(n), int(bool(is_complex)), out), "ssfm_device_reduce")
    return np.array(out[:], dtype=np.float64)


def prbs_device(order: int, length: int, seed: int, device: int = 0):
    """The reference's LFSR on the device: ``(bits DeviceArray uint8 (length,), final register state)``."""
    out = DeviceArray((int(length),), np.uint8, device)
    last = C.c_uint32(0)
    _check(load().ssfm_prbs(int(device), _VP(out.ptr), int(length), int(order), C.c_uint32(int(seed)), C.byref(last)), "ssfm_prbs")
    return out, int(last.value)


def real_device(a: DeviceArray) -> DeviceArray:
    out = DeviceArray(a.shape, np.float64, a.device)
    _check(load().ssfm_device_convert(a.device, _VP(a.ptr), C128, _VP(out.ptr), F64_REAL, a.size), "ssfm_device_convert")
    return out


def mzm_device(sig: DeviceArray, noise, drive: DeviceArray, drive_noise, k, bias, sqrt_loss, half_eta, dead_pol: int):
    """Mach-Zehnder transfer on device arrays (complex128 field(s), float64 or complex128 drive)."""
    n = sig.shape[-1]
    n_pol = 1 if sig.ndim == 1 else sig.shape[0]
    out_s = DeviceArray(sig.shape, np.complex128, sig.device)
    out_n = None if noise is None else DeviceArray(sig.shape, np.complex128, sig.device)
    p = lambda x: None if x is None else _VP(x.ptr)
    _check(load().ssfm_mzm(sig.device, _VP(out_s.ptr), p(out_n), _VP(sig.ptr), p(noise), n_pol, n, _VP(drive.ptr), p(drive_noise),
                           int(drive.dtype.kind == "c"), float(k), float(bias), float(sqrt_loss), float(half_eta), int(dead_pol)), "ssfm_mzm")
    return out_s, out_n


def cumsum_device(a: DeviceArray) -> DeviceArray:
    """``numpy.cumsum`` of a 1-D float64 device array."""
    out = DeviceArray(a.shape, np.float64, a.device)
    _check(load().ssfm_device_cumsum(a.device, _VP(out.ptr), _VP(a.ptr), a.size), "ssfm_device_cumsum")
    return out


def min_device(a: DeviceArray) -> float:
    m = C.c_double()
    _check(load().ssfm_device_reduce(a.device, REDUCE_MIN, _VP(a.ptr), None, 1, a.size, 0, C.byref(m)), "ssfm_device_reduce")
    return float(m.value)


def laser_device(n: int, amp: float, phase, rin, w, step: float, stop: float, device: int = 0) -> DeviceArray:
    """``ssfm_laser``: the CW field with optional phase noise / intensity noise (float64 DeviceArrays) / frequency offset ``w``."""
    out = DeviceArray((n,), np.complex128 if (phase is not None or w is not None) else np.float64, device)
    p = lambda x: None if x is None else _VP(x.ptr)
    _check(load().ssfm_laser(int(device), _VP(out.ptr), int(n), float(amp), p(phase), p(rin), int(w is not None), float(w or 0.0), float(step), float(stop)), "ssfm_laser")
    return out


def device_mem_info(device: int = 0):
    """(free, total, pooled) bytes of HBM on ``device``; ``pooled`` = freed buffers the library keeps for reuse."""
    f, t, p = C.c_size_t(), C.c_size_t(), C.c_size_t()
    _check(load().ssfm_device_mem_info(int(device), C.byref(f), C.byref(t), C.byref(p)), "ssfm_device_mem_info")
    return int(f.value), int(t.value), int(p.value)


def square_law_device(signal: DeviceArray, noise, r: float, post: float = 1.0):
    """:func:`square_law` on device-resident complex128 fields; returns float64 DeviceArrays (times ``post``)."""
    n = signal.shape[-1]
    n_pol = 1 if signal.ndim == 1 else signal.shape[0]
    i_sig = DeviceArray((n,), np.float64, signal.device)
    i_noise = None if noise is None else DeviceArray((n,), np.float64, signal.device)
    _check(load().ssfm_square_law(signal.device, _VP(signal.ptr), None if noise is None else _VP(noise.ptr), n_pol, n, float(r), float(post),
                                  _VP(i_sig.ptr), None if noise is None else _VP(i_noise.ptr), 1), "ssfm_square_law")
    return i_sig, i_noise


class Plan:
    """One propagator instance: device buffers + one HIP stream on one GPU."""

    def __init__(self, n: int, batch: int, precision: int = C64, device: int = 0):
        self._h = None
        lib = load()
        h = _VP()
        _check(lib.ssfm_plan_create(C.byref(h), int(device), int(n), int(batch), int(precision)), "ssfm_plan_create")
        self._h = h
        self.n, self.batch, self.precision, self.device = int(n), int(batch), int(precision), int(device)
        self.cdtype = _CDTYPE[precision]
        self.rdtype = _RDTYPE[precision]
        # A plan is ONE field buffer, one set of staging tables and one stream: a sequence such as set_field ...
        # propagate ... get_field must not interleave with another thread's on the same plan (ctypes releases the
        # GIL during calls).  The device functions hold this lock for their whole sequence; re-entrant because they nest.
        self.lock = threading.RLock()

    # what the staging buffers hold is the C plan's knowledge (ssfm_plan_set_tag / _get_tag): 0 operator, 1 / 2 table slots
    def tag(self, which: int) -> int:
        t = C.c_uint64(0)
        _check(load().ssfm_plan_get_tag(self._h, int(which), C.byref(t)), "ssfm_plan_get_tag")
        return int(t.value)

    def set_tag(self, which: int, tag: int):
        _check(load().ssfm_plan_set_tag(self._h, int(which), C.c_uint64(int(tag) & (2 ** 64 - 1))), "ssfm_plan_set_tag")

    @property
    def closed(self) -> bool:
        return self._h is None

    def close(self):
        if self._h is not None and _lib is not None:
            _lib.ssfm_plan_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- data movement
    def set_linear_operator(self, dtilde: np.ndarray):
        d = np.ascontiguousarray(dtilde, dtype=self.cdtype)
        if d.shape != (self.n,):
            raise ValueError(f"D~ must have shape ({self.n},), got {d.shape}")
        _check(load().ssfm_set_linear_operator(self._h, _ptr(d)), "ssfm_set_linear_operator")

    def set_field(self, field: np.ndarray):
        f = np.ascontiguousarray(field, dtype=self.cdtype).reshape(self.batch, self.n)
        _check(load().ssfm_set_field(self._h, _ptr(f), 0), "ssfm_set_field")

    def set_field_device(self, dev_ptr: int):
        _check(load().ssfm_set_field(self._h, C.c_void_p(dev_ptr), 1), "ssfm_set_field")

    def get_field(self) -> np.ndarray:
        out = host_empty((self.batch, self.n), self.cdtype)
        _check(load().ssfm_get_field(self._h, _ptr(out), 0), "ssfm_get_field")
        return out

    def get_field_device(self, dev_ptr: int):
        _check(load().ssfm_get_field(self._h, C.c_void_p(dev_ptr), 1), "ssfm_get_field")

    def copy_into_field(self, byte_offset: int, src_ptr: int, nbytes: int, on_device: bool):
        """Raw copy into the plan's field buffer at ``byte_offset`` (rows of several arrays side by side)."""
        _check(load().ssfm_synchronize(self._h), "ssfm_synchronize")
        _check(load().ssfm_device_copy(self.device, _VP(self.field_device_ptr + byte_offset), _VP(src_ptr), nbytes, 2 if on_device else 0),
               "ssfm_device_copy")
        TRANSFERS["h2d"] += 0 if on_device else 1

    def copy_from_field(self, byte_offset: int, dst_dev_ptr: int, nbytes: int):
        """Raw device-to-device copy out of the plan's field buffer."""
        _check(load().ssfm_synchronize(self._h), "ssfm_synchronize")
        _check(load().ssfm_device_copy(self.device, _VP(dst_dev_ptr), _VP(self.field_device_ptr + byte_offset), nbytes, 2), "ssfm_device_copy")

    @property
    def field_device_ptr(self) -> int:
        return int(load().ssfm_field_device_ptr(self._h) or 0)

    @property
    def stream(self) -> int:
        return int(load().ssfm_stream(self._h) or 0)

    # -- compute
    def propagate_fixed(self, gamma: float, h_schedule, snapshots: bool = False):
        hs = np.ascontiguousarray(h_schedule, dtype=self.rdtype)
        snap = None
        if snapshots:
            snap = host_empty((hs.size + 1, self.batch, self.n), self.cdtype)
        _check(load().ssfm_propagate_fixed(self._h, float(gamma), _ptr(hs), hs.size,
                                           _ptr(snap) if snap is not None else None), "ssfm_propagate_fixed")
        return snap

    def propagate_fixed_capture(self, gamma: float, h_schedule, every=None, scalars: bool = False) -> dict:
        """A fixed-step run with a z-resolved capture that does not stall it (``ssfm_propagate_fixed_capture``): the field after every ``every``-th step
        (and the input, and the last step) into page-locked memory while the run goes on, and / or the per-step scalars.  Returns a dict with
        ``steps`` (indices of the captured steps, 0 = the input), ``fields`` (len(steps), batch, n) and, with ``scalars``, ``power`` and ``peak``
        (nsteps + 1, batch): mean and maximum of |A|^2 of every row after every step.  Synchronous at this level (the C call is not)."""
        hs = np.ascontiguousarray(h_schedule, dtype=self.rdtype)
        if hs.size < 1:
            raise ValueError("the schedule is empty")
        out, cap = {}, Capture(0, None, None)
        if every is not None:
            every = int(every)
            if every < 1:
                raise ValueError(f"every = {every}")
            idx = list(range(0, hs.size, every)) + [hs.size]
            out["steps"] = np.asarray(idx, dtype=np.int64)
            # (page-locked up to 8 GiB: the transfers of a strided capture run beside the kernels only into page-locked memory)
            out["fields"] = host_empty((len(idx), self.batch, self.n), self.cdtype, limit=8 << 30)
            cap.every, cap.fields = every, out["fields"].ctypes.data
        raw = None
        if scalars:
            raw = host_empty((hs.size + 1, self.batch, 2), np.float64)
            cap.scalars = raw.ctypes.data
        _check(load().ssfm_propagate_fixed_capture(self._h, float(gamma), _ptr(hs), hs.size, C.byref(cap)), "ssfm_propagate_fixed_capture")
        self.synchronize()
        if raw is not None:
            out["power"], out["peak"] = raw[..., 0].copy(), raw[..., 1].copy()
        return out

    def propagate_adaptive(self, gamma, length, phi_max, single_step, max_steps=1 << 16, snapshots=False):
        """Adaptive run; returns ``(steps, z float64 (steps + 1,), snapshots or None)``.  A z-resolved capture is taken in
        blocks of at most 256 MiB of page-locked memory (``ssfm_adaptive_run``), so its memory follows the steps actually
        taken, not ``max_steps``."""
        lib = load()
        steps, done = _I64(0), _I(0)
        if not snapshots:
            _check(lib.ssfm_adaptive_begin(self._h, float(gamma), float(length), float(phi_max), int(bool(single_step)), int(max_steps), 0), "ssfm_adaptive_begin")
            _check(lib.ssfm_adaptive_run(self._h, int(max_steps), None, C.byref(steps), C.byref(done)), "ssfm_adaptive_run")
            blocks = None
        else:
            blocks = [self.get_field().reshape(1, self.batch, self.n)]                      # the input (devices.py:1150-1152)
            _check(lib.ssfm_adaptive_begin(self._h, float(gamma), float(length), float(phi_max), int(bool(single_step)), int(max_steps), 1), "ssfm_adaptive_begin")
            per = max(1, min(64, (256 << 20) // max(1, self.batch * self.n * np.dtype(self.cdtype).itemsize)))
            while not done.value and steps.value < max_steps:
                before = steps.value
                blk = host_empty((per, self.batch, self.n), self.cdtype)
                _check(lib.ssfm_adaptive_run(self._h, per, _ptr(blk), C.byref(steps), C.byref(done)), "ssfm_adaptive_run")
                blocks.append(blk[: steps.value - before])
        z = np.zeros(steps.value + 1, dtype=np.float64)
        _check(lib.ssfm_adaptive_finish(self._h, C.byref(steps), z.ctypes.data_as(C.POINTER(_D))), "ssfm_adaptive_finish")
        s = steps.value
        snap = None
        if blocks is not None:
            snap = blocks[0] if len(blocks) == 1 else np.concatenate(blocks, axis=0)       # a compact copy: nothing of the blocks stays alive
        return s, z, snap

    def propagate_adaptive_capture(self, gamma, length, phi_max, every=None, steps=None, capacity=None, max_steps=1 << 16, fields=None):
        """An adaptive run with a z-resolved capture that keeps the run's engine (``ssfm_adaptive_set_capture``): the field after every ``every``-th step, or
        after the (1-based, ascending) step numbers ``steps``.  Returns ``(n_steps, z float64 (n_steps + 1,), taken int64 (k,), fields (k, batch, n))`` --
        the input and the end field are not among the snapshots (``get_field`` before and after).  ``fields``: a destination made earlier
        (``host_empty((capacity, batch, n), cdtype)``: page-locking a GiB takes longer than the run)."""
        lib = load()
        if (every is None) == (steps is None):
            raise ValueError("either `every` or `steps`")
        if steps is not None:
            want = np.ascontiguousarray(steps, dtype=np.int64)
            if want.size == 0:
                raise ValueError("`steps` is empty")
            capacity = want.size if capacity is None else int(capacity)
        else:
            every = int(every)
            if every < 1:
                raise ValueError(f"every = {every}")
            want = None
            if capacity is None:
                raise ValueError("a strided capture of an adaptive run needs a `capacity` (snapshots): the run's length in steps is its own")
        capacity = max(1, int(capacity))
        if fields is None:
            fields = host_empty((capacity, self.batch, self.n), self.cdtype, limit=8 << 30)
        elif fields.shape != (capacity, self.batch, self.n) or fields.dtype != self.cdtype or not fields.flags.c_contiguous:
            raise ValueError("`fields` must be a C-contiguous (capacity, batch, n) array of the plan's type")
        taken = np.zeros(capacity, dtype=np.int64)
        n_taken = C.c_int64(0)
        cap = AdaptiveCapture(every if want is None else 0, want.ctypes.data_as(C.POINTER(C.c_int64)) if want is not None else None, 0 if want is None else want.size,
                              fields.ctypes.data, capacity, taken.ctypes.data_as(C.POINTER(C.c_int64)), C.pointer(n_taken))
        nsteps, done = _I64(0), _I(0)
        _check(lib.ssfm_adaptive_begin(self._h, float(gamma), float(length), float(phi_max), 0, int(max_steps), 0), "ssfm_adaptive_begin")
        _check(lib.ssfm_adaptive_set_capture(self._h, C.byref(cap)), "ssfm_adaptive_set_capture")
        _check(lib.ssfm_adaptive_run(self._h, int(max_steps), None, C.byref(nsteps), C.byref(done)), "ssfm_adaptive_run")
        z = np.zeros(nsteps.value + 1, dtype=np.float64)
        _check(lib.ssfm_adaptive_finish(self._h, C.byref(nsteps), z.ctypes.data_as(C.POINTER(_D))), "ssfm_adaptive_finish")
        k = int(n_taken.value)
        return nsteps.value, z, taken[:k].copy(), fields[:k]

    def apply_transfer(self, H: np.ndarray):
        h = np.ascontiguousarray(H, dtype=self.cdtype)
        if h.shape != (self.n,):
            raise ValueError(f"H must have shape ({self.n},), got {h.shape}")
        _check(load().ssfm_apply_transfer(self._h, _ptr(h)), "ssfm_apply_transfer")

    def apply_dispersion(self, dt: float, D_s2: float, want_H: bool = False):
        """DM with H generated on the device; returns natural-order H if ``want_H``."""
        H = np.empty(self.n, dtype=self.cdtype) if want_H else None
        _check(load().ssfm_apply_dispersion(self._h, float(dt), float(D_s2), _ptr(H) if want_H else None), "ssfm_apply_dispersion")
        return H

    # -- building blocks of the chirp-z path (lengths that are not powers of two)
    def transfer_table(self, H: np.ndarray, slot: int):
        h = np.ascontiguousarray(H, dtype=self.cdtype)
        if h.shape != (self.n,):
            raise ValueError(f"H must have shape ({self.n},), got {h.shape}")
        _check(load().ssfm_transfer_table(self._h, _ptr(h), int(slot)), "ssfm_transfer_table")

    def apply_table(self, slot: int):
        _check(load().ssfm_apply_table(self._h, int(slot)), "ssfm_apply_table")

    def load_padded(self, src: "DeviceArray"):
        """field <- the float64 / complex128 device array ``src``, zero-padded to the plan length (batch 1)."""
        _check(load().ssfm_load_padded(self._h, self.n, _VP(src.ptr), int(src.dtype.kind == "c"), src.size), "ssfm_load_padded")

    def load_symbols(self, sym: "DeviceArray", up: int):
        """field <- the float64 amplitudes ``sym`` zero-stuffed to ``up`` samples per symbol (sample at ``up // 2``)."""
        _check(load().ssfm_load_symbols(self._h, self.n, _VP(sym.ptr), 0, sym.size, int(up)), "ssfm_load_symbols")

    def load_bits(self, bits: "DeviceArray", up: int):
        """field <- device-resident bits (uint8) as amplitudes 0.0 / 1.0, zero-stuffed to ``up`` samples per bit."""
        _check(load().ssfm_load_symbols(self._h, self.n, _VP(bits.ptr), 1, bits.size, int(up)), "ssfm_load_symbols")

    def load_qpsk(self, bits: "DeviceArray", nsym: int, sps: int):
        """field rows <- QPSK-like symbols from device-resident bits (``ssfm_load_qpsk``)."""
        _check(load().ssfm_load_qpsk(self._h, self.n, self.batch, _VP(bits.ptr), int(nsym), int(sps)), "ssfm_load_qpsk")

    def chirp_setup(self, n: int):
        """Slots 0 / 1 <- the transfer functions of Bluestein's two convolutions for fields of ``n`` samples, generated and transformed on the device
        (``ssfm_chirp_setup``; the plan's field is consumed)."""
        _check(load().ssfm_chirp_setup(self._h, self.n, int(n)), "ssfm_chirp_setup")

    def load_pulse(self, kind: int, npts: int, start: float, step: float, stop: float, pow2m: int, params):
        """field <- one of the DAC's built-in pulses over ``linspace(start, stop, npts)``, zero-padded (``ssfm_load_pulse``)."""
        p = (C.c_double * 7)(*(list(params) + [0.0] * (7 - len(params))))
        _check(load().ssfm_load_pulse(self._h, self.n, int(kind), int(npts), float(start), float(step), float(stop), int(pow2m), p), "ssfm_load_pulse")

    def table_from_field(self, slot: int):
        _check(load().ssfm_table_from_field(self._h, int(slot)), "ssfm_table_from_field")

    def chirp_transfer(self, A: "DeviceArray", chirp: "DeviceArray", tab: "DeviceArray", exponent: bool):
        """Rows of ``A`` (batch, n complex128) <- ifft(fft(row) * tab), or * exp(tab) with ``exponent`` (``ssfm_chirp_transfer``); asynchronous."""
        _check(load().ssfm_chirp_transfer(self._h, self.n, self.batch, _VP(A.ptr), _VP(chirp.ptr), _VP(tab.ptr), A.shape[-1], 1 if exponent else 0), "ssfm_chirp_transfer")

    def chirp_fourier(self, A: "DeviceArray", chirp: "DeviceArray", chirp_conj: "DeviceArray", inverse: bool):
        """Rows of ``A`` (batch, n complex128) <- numpy.fft.fft / ifft of the row (``ssfm_chirp_fourier``); asynchronous."""
        _check(load().ssfm_chirp_fourier(self._h, self.n, self.batch, _VP(A.ptr), _VP(chirp.ptr), _VP(chirp_conj.ptr), A.shape[-1], 1 if inverse else 0), "ssfm_chirp_fourier")

    def chirp_propagate(self, A: "DeviceArray", P: "DeviceArray", chirp: "DeviceArray", Dt: "DeviceArray", gamma: float, hs=None, *, length: float = 0.0,
                        phi_max: float = 0.0, f32: bool = True, max_steps: int = 1 << 20):
        """A whole chirp-z run from C (ssfm_chirp_propagate).  ``hs``: float64 host array of step sizes (fixed step) or None (adaptive);
        returns (steps, z) with z = positions after every step (adaptive) or None (fixed)."""
        steps = _I64(0)
        if hs is not None:
            hs = np.ascontiguousarray(hs, dtype=np.float64)
            if hs.size == 0:                       # (length 0: the reference's loop does not run, devices.py:1172)
                return 0, None
            _check(load().ssfm_chirp_propagate(self._h, self.n, self.batch, _VP(A.ptr), _VP(P.ptr), _VP(chirp.ptr), _VP(Dt.ptr), A.shape[-1], float(gamma),
                                               hs.ctypes.data_as(C.POINTER(_D)), hs.size, 0.0, 0.0, 1 if f32 else 0, 1, None, C.byref(steps)), "ssfm_chirp_propagate")
            return int(steps.value), None
        z = np.zeros(int(max_steps) + 1, dtype=np.float64)
        _check(load().ssfm_chirp_propagate(self._h, self.n, self.batch, _VP(A.ptr), _VP(P.ptr), _VP(chirp.ptr), _VP(Dt.ptr), A.shape[-1], float(gamma),
                                           None, 0, float(length), float(phi_max), 1 if f32 else 0, int(max_steps), z.ctypes.data_as(C.POINTER(_D)), C.byref(steps)),
               "ssfm_chirp_propagate")
        return int(steps.value), z[: int(steps.value) + 1]

    def chirp_propagate_c64(self, A: "DeviceArray", chirp: "DeviceArray", Dt: "DeviceArray", gamma: float, hs=None, *, length: float = 0.0, phi_max: float = 0.0,
                            max_steps: int = 1 << 17):
        """A chirp-z run of ``A`` (batch, n <= plan length / 2; complex64) in ONE launch on this complex64 plan's line (``ssfm_chirp_propagate_c64``): a
        workgroup per row up to 4096 points of line, the one-XCD engine above.  ``hs``: float64 host array of step sizes (fixed step; a step of length zero is
        the identity) -> True, or None for the adaptive rule over ``length`` -> (steps, z).  False / None: the plan has no such engine or schedule, or the
        launch's workgroups did not meet -- ``A`` is then as it was."""
        steps = _I64(0)
        if hs is not None:
            hs = np.ascontiguousarray(hs, dtype=np.float64)
            hs = hs[hs != 0.0]
            if hs.size == 0:
                return True
            rc = load().ssfm_chirp_propagate_c64(self._h, _VP(A.ptr), _VP(chirp.ptr), _VP(Dt.ptr), A.shape[-1], float(gamma), hs.ctypes.data_as(C.POINTER(_D)), hs.size,
                                                 0.0, 0.0, 1, None, C.byref(steps))
            if rc == 2:
                return False
            _check(rc, "ssfm_chirp_propagate_c64")
            return True
        z = np.zeros(int(max_steps) + 1, np.float64)
        rc = load().ssfm_chirp_propagate_c64(self._h, _VP(A.ptr), _VP(chirp.ptr), _VP(Dt.ptr), A.shape[-1], float(gamma), None, 0, float(length), float(phi_max),
                                             int(max_steps), z.ctypes.data_as(C.POINTER(_D)), C.byref(steps))
        if rc == 2:
            return None
        _check(rc, "ssfm_chirp_propagate_c64")
        return int(steps.value), z[: int(steps.value) + 1]

    def debug_fft(self) -> np.ndarray:
        out = host_empty((self.batch, self.n), self.cdtype)
        _check(load().ssfm_debug(self._h, 0, 0, _ptr(out)), "ssfm_debug")
        return out

    def synchronize(self):
        _check(load().ssfm_synchronize(self._h), "ssfm_synchronize")

    def set_profiling(self, mode: int):
        """0 off, 1 event per launch (per-class times, perturbs), 2 event per 64 launches (pooled, cheap)."""
        _check(load().ssfm_set_profiling(self._h, int(mode)), "ssfm_set_profiling")

    @property
    def lanes(self) -> int:
        """Row groups a fixed-step run drives on separate streams, as configured at creation (``ssfm_run_info.lanes_configured``)."""
        r = RunInfo()
        _check(load().ssfm_last_run_info(self._h, C.byref(r), C.sizeof(r)), "ssfm_last_run_info")
        return int(r.lanes_configured)

    def kernel_times(self):
        """{'k_time': (launches, total_ms), 'k_freq': (launches, total_ms)} of the last profiled run."""
        cnt = (_I64 * 2)()
        ms = (_D * 2)()
        _check(load().ssfm_kernel_times(self._h, cnt, ms), "ssfm_kernel_times")
        return {"k_time": (cnt[0], ms[0]), "k_freq": (cnt[1], ms[1])}

    def last_run_info(self) -> dict:
        """Which engine the last run took (``ENGINES``), whether it had to be repeated on a fallback, such repeats over the plan's life,
        and whether the plan's lanes share a hardware queue (include/ssfm_amd.h ``ssfm_last_run_info``).  Synchronises first: a one-launch
        run of a medium plan knows at its end whether its workgroups met."""
        _check(load().ssfm_synchronize(self._h), "ssfm_synchronize")
        r = RunInfo()
        _check(load().ssfm_last_run_info(self._h, C.byref(r), C.sizeof(r)), "ssfm_last_run_info")
        return {"engine": ENGINES[r.engine] if 0 <= r.engine < len(ENGINES) else str(r.engine), "fell_back": bool(r.fell_back),
                "fallbacks_total": int(r.fallbacks_total), "lanes": int(r.lanes), "lanes_share_queue": bool(r.lanes_share_queue),
                "lanes_remade": int(r.lanes_remade), "lanes_dropped": bool(r.lanes_dropped), "lane_heals": int(r.lane_heals), "lane_alone_us": float(r.lane_alone_us),
                "lane_pair_us": float(r.lane_pair_us), "lane_last_us": float(r.lane_last_us), "lane_score": float(r.lane_score),
                "lanes_from_pool": bool(r.lanes_from_pool), "lane_ratings_total": int(r.lane_ratings_total), "lane_pairs_reused": int(r.lane_pairs_reused)}

    def lane_fault(self, mode: int):
        """Test hook of the lane health check (include/ssfm_amd.h ``ssfm_debug``, SSFM_DEBUG_LANE_FAULT)."""
        _check(load().ssfm_debug(self._h, 1, int(mode), None), "ssfm_debug")

    def last_propagate_ms(self):
        ms, n = C.c_float(0), _I64(0)
        _check(load().ssfm_last_propagate_ms(self._h, C.byref(ms), C.byref(n)), "ssfm_last_propagate_ms")
        return ms.value, n.value
