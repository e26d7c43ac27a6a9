"""Boundary types of the fibre path: the slice of ``opticomlib.typing`` that ``FIBER`` /
``DBP`` / ``DM`` touch, and nothing else.

* :data:`NULL`      -- "no noise" sentinel (reference ``typing.py:56-93``): ``x + NULL == x``.
* :data:`gv`        -- global sampling parameters; a signal does not carry its sample rate,
                       ``optical_signal.w()`` reads ``gv.dt`` at call time (``typing.py:1641``).
* :class:`optical_signal` -- ``.signal``, ``.noise``, ``.n_pol``, ``.size``, ``.execution_time``,
                       ``.w()``, ``.to_numpy()`` with the reference's shape -> ``n_pol`` rules
                       (``typing.py:2124-2196``).

Operators, plotting, PSD, eye diagrams etc. are out of scope.
"""
from __future__ import annotations

import numpy as np


_C_LIGHT = 299792458.0       # scipy.constants.c


class _NullType:
    """Additive identity that swallows everything else."""

    _inst = None

    def __new__(cls):
        if cls._inst is None:
            cls._inst = super().__new__(cls)
        return cls._inst

    def __repr__(self):
        return "NULL"

    def __add__(self, other):
        return other

    __radd__ = __add__

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method == "__call__" and ufunc in (np.add, np.subtract) and inputs[1] is self:
            return inputs[0]
        return self

    def __bool__(self):
        return False


NULL = _NullType()


class _GlobalVars:
    """Sampling grid shared by all signals (reference ``typing.py:106-388``, the part the path
    needs: ``sps``, ``R``, ``fs``, ``dt``, and ``wavelength`` / ``f0`` for the EDFA's ASE power)."""

    def __init__(self):
        self.default()

    def default(self):
        self.sps = 16
        self.R = 1e9
        self.fs = self.R * self.sps
        self.dt = 1 / self.fs
        self.wavelength = 1550e-9              # reference typing.py:207-209
        self.f0 = _C_LIGHT / self.wavelength
        self.N = 128                           # number of bit slots (typing.py:211); only LASER reads it, through `t`
        self.t = np.linspace(0, self.N * self.sps * self.dt, self.N * self.sps, endpoint=True)      # typing.py:213
        return self

    def __call__(self, sps=None, R=None, fs=None, wavelength=1550e-9, N=None, **extra):
        # same precedence as the reference (typing.py:306-335)
        if sps:
            self.sps = int(np.round(sps))
            if R:
                self.R = R
                self.fs = R * self.sps
            elif fs:
                self.fs = fs
                self.R = fs / self.sps
            else:
                self.fs = self.R * self.sps
        elif R:
            self.R = R
            if fs:
                self.fs = fs
                self.sps = int(np.round(fs / R))
            else:
                self.fs = R * self.sps
        elif fs:
            self.fs = fs
            self.sps = int(np.round(fs / self.R))
        self.dt = 1 / self.fs
        self.N = N if N is not None else self.N
        self.t = np.linspace(0, self.N * self.sps / self.fs, self.N * self.sps, endpoint=True)      # typing.py:357
        self.wavelength = wavelength           # reset to the default on every call, like the reference (typing.py:340-341)
        self.f0 = _C_LIGHT / wavelength
        for k, v in extra.items():
            setattr(self, k, v)
        return self


gv = _GlobalVars()


def _is_device(a) -> bool:
    """A ``_lib.DeviceArray`` (checked by duck typing so that this module never imports the HIP binding)."""
    return hasattr(a, "to_host") and hasattr(a, "ptr")


class _LazyArray:
    """``signal`` / ``noise`` attribute that may be backed by a device-resident array.

    Device calls hand their results over as ``DeviceArray``; the first host access downloads the data and
    the object is an ordinary NumPy-backed signal from then on (a host array may be modified in place, so
    the device copy is dropped rather than kept in sync).  Device-aware callers read ``obj._raw(name)``
    instead, which never transfers anything."""

    def __set_name__(self, owner, name):
        self.slot = "_" + name

    def __get__(self, obj, objtype=None):
        if obj is None:
            return self
        a = obj.__dict__.get(self.slot, NULL)
        if _is_device(a):
            a = a.to_host()
            obj.__dict__[self.slot] = a
        return a

    def __set__(self, obj, value):
        obj.__dict__[self.slot] = value


class binary_sequence:
    """Bit sequence container: the slice of reference ``typing.py:402-1020`` that ``PRBS`` returns and a DAC
    consumes -- ``.data`` (uint8 0/1), ``.size``, ``len()``, ``.to_numpy()``, comparison with array-likes.
    ``PRBS`` leaves its bits in GPU memory (``from_device``); ``.data`` downloads them on first access."""

    data = _LazyArray()

    @classmethod
    def from_device(cls, bits):
        """Wrap a device-resident uint8 array of 0 / 1 values without copying it to the host."""
        self = cls.__new__(cls)
        if bits.ndim != 1 or np.dtype(bits.dtype) != np.uint8:
            raise ValueError(f"Binary sequence must be a 1D uint8 array, got {bits.dtype} {bits.shape}")
        self.data = bits
        self.execution_time = 0.0
        return self

    def _raw(self):
        return self.__dict__.get("_data")

    def __init__(self, data):
        if isinstance(data, binary_sequence):
            data = data.data
        if isinstance(data, str):
            data = [int(ch) for ch in data.replace(" ", "").replace(",", "")]
        d = np.asarray(data)
        if d.ndim == 0:
            d = d[np.newaxis]
        if d.ndim != 1:
            raise ValueError(f"Binary sequence must be 1D, invalid shape {d.shape}")
        if not np.all((d == 0) | (d == 1)):
            raise ValueError("Binary sequence must contain only 0 and 1 values.")
        self.data = d.astype(np.uint8)
        self.execution_time = 0.0

    @property
    def size(self) -> int:
        return int(self._raw().size)

    @property
    def type(self):
        return binary_sequence

    @property
    def ones(self) -> int:
        return int(self.data.sum())

    @property
    def zeros(self) -> int:
        return self.size - self.ones

    def __len__(self):
        return self.size

    def __getitem__(self, key):
        r = self.data[key]
        return binary_sequence(r) if isinstance(r, np.ndarray) else int(r)

    def __eq__(self, other):
        other = other.data if isinstance(other, binary_sequence) else np.asarray(other)
        return self.data == other

    def __array__(self, dtype=None, copy=None):
        return self.data if dtype is None else self.data.astype(dtype)

    def to_numpy(self) -> np.ndarray:
        return self.data

    def __repr__(self):
        if _is_device(self._raw()):
            return f"binary_sequence(size={self.size}, on GPU {self._raw().device})"
        return f"binary_sequence({np.array2string(self.data, threshold=20)})"


class electrical_signal:
    """1-D electrical signal with optional noise (the slice ``LPF`` needs of reference
    ``typing.py:1022-1165``)."""

    signal = _LazyArray()
    noise = _LazyArray()

    @classmethod
    def from_device(cls, signal, noise=NULL):
        """Wrap device-resident 1-D arrays without copying them to the host."""
        self = cls.__new__(cls)
        if signal.ndim != 1 or (noise is not NULL and noise.shape != signal.shape):
            raise ValueError(f"Signal must be 1D array for electrical_signal, invalid shape {signal.shape}")
        self.signal, self.noise = signal, noise
        self.execution_time = 0.0
        return self

    def _raw(self, name):
        return self.__dict__.get("_" + name, NULL)

    def __init__(self, signal, noise=NULL, dtype=None):
        if isinstance(signal, electrical_signal):
            noise = signal.noise if noise is NULL else np.asarray(noise) + signal.noise
            signal = signal.signal
        sig = np.asarray(signal)
        noi = noise
        if noi is not NULL:
            noi = np.asarray(noi)
            common = np.result_type(sig, noi) if dtype is None else dtype
            sig, noi = sig.astype(common, copy=False), noi.astype(common, copy=False)
            if sig.shape != noi.shape:
                raise ValueError(f"`signal` and `noise` must have the same shape, mismatch shapes {sig.shape} and {noi.shape}!")
        elif dtype is not None:
            sig = sig.astype(dtype, copy=False)
        if sig.ndim > 1 or sig.size < 1:
            raise ValueError(f"Signal must be scalar or 1D array for electrical_signal, invalid shape {sig.shape}")
        if sig.ndim == 0:
            sig = sig[np.newaxis]
            if noi is not NULL:
                noi = noi[np.newaxis]
        self.signal = sig
        self.noise = noi
        self.execution_time = 0.0

    @property
    def size(self) -> int:
        return int(self._raw("signal").size)

    @property
    def ndim(self) -> int:
        return self._raw("signal").ndim

    def __len__(self):
        return self.size

    def __call__(self, domain, shift: bool = False):
        """New object holding the FFT (``'w'`` / ``'f'``) or inverse FFT (``'t'``) of signal and noise along the last axis
        (reference ``typing.py:1421-1462``); ``shift`` applies fftshift / ifftshift.  Computed on the GPU."""
        from . import devices
        return devices._fourier(self, domain, shift)

    def to_numpy(self) -> np.ndarray:
        return np.asarray(self.signal + self.noise)

    def __repr__(self):
        return f"electrical_signal(size={self.size}, dtype={self._raw('signal').dtype}, noise={'NULL' if self._raw('noise') is NULL else 'array'})"


class optical_signal:
    """Optical field container: ``signal`` (and optional ``noise``) of shape ``(N,)`` for one
    polarisation or ``(2, N)`` for two."""

    signal = _LazyArray()
    noise = _LazyArray()

    @classmethod
    def from_device(cls, signal, noise=NULL, n_pol=None):
        """Wrap device-resident arrays of shape ``(N,)`` or ``(2, N)`` without copying them to the host."""
        self = cls.__new__(cls)
        if signal.ndim not in (1, 2) or (signal.ndim == 2 and signal.shape[0] != 2):
            raise ValueError(f"device-resident optical_signal needs shape (N,) or (2, N), got {signal.shape}")
        if noise is not NULL and tuple(noise.shape) != tuple(signal.shape):
            raise ValueError(f"`signal` and `noise` must have the same shape, mismatch shapes {signal.shape} and {noise.shape}!")
        self.signal, self.noise = signal, noise
        self.n_pol = signal.ndim if n_pol is None else n_pol
        self.execution_time = 0.0
        return self

    def _raw(self, name):
        return self.__dict__.get("_" + name, NULL)

    @property
    def on_device(self) -> bool:
        """True while ``signal`` still lives in GPU memory only (no host copy has been asked for)."""
        return _is_device(self._raw("signal"))

    def __init__(self, signal, noise=NULL, n_pol=None, dtype=None):
        if _is_device(signal) and (noise is NULL or _is_device(noise)) and dtype is None:      # device arrays: no copy to the host
            other = optical_signal.from_device(signal, noise, n_pol)
            self.__dict__.update(other.__dict__)
            return
        if _is_device(signal):
            signal = signal.to_host()
        if _is_device(noise):
            noise = noise.to_host()
        if isinstance(signal, optical_signal):
            if noise is NULL and n_pol in (None, signal.n_pol) and dtype is None:      # plain re-wrap: keep device residency
                self.signal, self.noise, self.n_pol, self.execution_time = signal._raw("signal"), signal._raw("noise"), signal.n_pol, 0.0
                return
            if noise is not NULL:
                noise = np.asarray(noise) + signal.noise
            else:
                noise = signal.noise
            signal = signal.signal
        sig = np.asarray(signal)
        noi = noise
        if noi is not NULL:
            noi = np.asarray(noi)
            common = np.result_type(sig, noi) if dtype is None else dtype
            sig, noi = sig.astype(common, copy=False), noi.astype(common, copy=False)
            if sig.shape != noi.shape:
                raise ValueError(f"`signal` and `noise` must have the same shape, mismatch shapes {sig.shape} and {noi.shape}!")
        elif dtype is not None:
            sig = sig.astype(dtype, copy=False)

        if sig.ndim > 2 or (sig.ndim > 1 and sig.shape[0] > 2) or sig.size < 1:
            raise ValueError(f"Signal must be a scalar, 1D or 2D array for optical_signal, invalid shape {sig.shape}")
        if n_pol is not None and n_pol not in (1, 2):
            raise ValueError("n_pol must be either 1 or 2")

        def both(f):
            return f(sig), (noi if noi is NULL else f(noi))

        want_two = n_pol == 2
        if sig.ndim == 0:
            if want_two:
                sig, noi = both(lambda a: np.array([[a], [a]]))
            else:
                sig, noi = both(lambda a: a[np.newaxis])
                n_pol = 1
        elif sig.ndim == 1:
            if want_two:
                sig, noi = both(lambda a: np.array([a, a]))
            else:
                n_pol = 1
        elif sig.shape[0] == 1:            # (1, N): becomes dual-pol by tiling unless n_pol=1
            if n_pol in (None, 2):
                sig, noi = both(lambda a: np.tile(a, (2, 1)))
                n_pol = 2
            else:
                sig, noi = both(lambda a: a[0])
        else:                              # (2, N)
            if n_pol in (None, 2):
                n_pol = 2
            else:
                sig, noi = both(lambda a: a[0])

        self.signal = sig
        self.noise = noi
        self.n_pol = n_pol
        self.execution_time = 0.0

    @property
    def size(self) -> int:
        """Samples per polarisation (reference ``typing.py:2313-2320``)."""
        return int(self._raw("signal").shape[-1])

    def __len__(self):
        return self.size

    @property
    def shape(self):
        return tuple(self._raw("signal").shape)

    @property
    def dt(self):
        return gv.dt

    def __call__(self, domain, shift: bool = False):
        """New object holding the FFT (``'w'`` / ``'f'``) or inverse FFT (``'t'``) of signal and noise along the last axis
        (reference ``typing.py:1421-1462``); ``shift`` applies fftshift / ifftshift.  Computed on the GPU."""
        from . import devices
        return devices._fourier(self, domain, shift)

    def to_numpy(self) -> np.ndarray:
        """``signal + noise`` (reference ``typing.py:1593-1597``)."""
        return np.asarray(self.signal + self.noise)

    def w(self, shift: bool = False) -> np.ndarray:
        """Angular frequency grid [rad/s], FFT order (reference ``typing.py:1628-1644``)."""
        w = np.fft.fftfreq(self.size, gv.dt) * 2 * np.pi
        return np.fft.fftshift(w, axes=-1) if shift else w

    # -- the little signal algebra a link script uses around the devices (reference typing.py:1308-1344, :1599-1608,
    #    :1663-1720); host arithmetic on materialised arrays
    def abs(self, of: str = "all") -> np.ndarray:
        """``|signal|``, ``|noise|`` (zeros without noise) or ``|signal + noise|``."""
        of = of.lower()
        if of == "signal":
            return np.abs(self.signal)
        if of == "noise":
            return np.zeros_like(np.real(self.signal)) if self.noise is NULL else np.abs(self.noise)
        if of == "all":
            return np.abs(self.to_numpy())
        raise ValueError('`of` must be one of the following values ("signal", "noise", "all")')

    def power(self, unit: str = "W", of: str = "all"):
        """Mean power per polarisation of the signal, the noise or both, in W or dBm."""
        p = np.mean(self.abs(of) ** 2, axis=-1)
        unit = unit.lower()
        if unit == "w":
            return p
        if unit == "dbm":
            return 10 * np.log10(p) + 30
        raise ValueError('`unit` must be one of the following values ("W", "dBm")')

    def phase(self) -> np.ndarray:
        return np.unwrap(np.angle(self.to_numpy()))

    def conj(self):
        return optical_signal(np.conj(self.signal), NULL if self.noise is NULL else np.conj(self.noise), n_pol=self.n_pol)

    def _other(self, other):
        return (other.signal, other.noise) if isinstance(other, (optical_signal, electrical_signal)) else (np.asarray(other), NULL)

    def _device_pair(self, other):
        """Both operands entirely in the memory of one GPU, same shape and complex type: the sum can stay there."""
        if not isinstance(other, optical_signal):
            return False
        a, b = self._raw("signal"), other._raw("signal")
        if not (_is_device(a) and _is_device(b)) or tuple(a.shape) != tuple(b.shape) or a.dtype != b.dtype or a.dtype.kind != "c" or a.device != b.device:
            return False
        return all(x is NULL or (_is_device(x) and x.dtype == a.dtype and x.device == a.device) for x in (self._raw("noise"), other._raw("noise")))

    def __add__(self, other):
        if self._device_pair(other):                          # e.g. two WDM channels coming out of their modulators
            n1, n2 = self._raw("noise"), other._raw("noise")
            noise = n2 if n1 is NULL else (n1 if n2 is NULL else n1 + n2)
            return optical_signal.from_device(self._raw("signal") + other._raw("signal"), noise, n_pol=self.n_pol)
        s, n = self._other(other)
        return optical_signal(self.signal + s, self.noise + n, n_pol=self.n_pol)

    __radd__ = __add__

    def __neg__(self):
        return optical_signal(-self.signal, NULL if self.noise is NULL else -self.noise, n_pol=self.n_pol)

    def __sub__(self, other):
        s, n = self._other(other)
        return optical_signal(self.signal - s, self.noise - n if n is not NULL else self.noise, n_pol=self.n_pol)

    def __mul__(self, other):
        """``(s1 + n1)(s2 + n2)``: the signal is ``s1 s2``, everything that contains a noise factor is noise."""
        raw_s, raw_n = self._raw("signal"), self._raw("noise")
        if isinstance(other, (int, float)) and not isinstance(other, bool) and _is_device(raw_s) and raw_s.dtype in (np.complex128, np.float64) \
                and (raw_n is NULL or (_is_device(raw_n) and raw_n.dtype == raw_s.dtype)):
            from . import _lib                                 # a real gain / loss factor on a device-resident signal
            return optical_signal.from_device(_lib.scale_add_device(raw_s, float(other)),
                                              NULL if raw_n is NULL else _lib.scale_add_device(raw_n, float(other)), n_pol=self.n_pol)
        s, n = self._other(other)
        sig = self.signal * s
        noi = NULL
        for term in ((self.signal * n) if n is not NULL else NULL, (self.noise * s) if self.noise is not NULL else NULL,
                     (self.noise * n) if (self.noise is not NULL and n is not NULL) else NULL):
            noi = noi + term
        return optical_signal(sig, noi, n_pol=self.n_pol)

    __rmul__ = __mul__

    def __getitem__(self, key):
        return optical_signal(self.signal[..., key] if self.n_pol == 2 else self.signal[key],
                              NULL if self.noise is NULL else (self.noise[..., key] if self.n_pol == 2 else self.noise[key]), n_pol=self.n_pol)

    def __repr__(self):
        where = " [device]" if self.on_device else ""
        return f"optical_signal(n_pol={self.n_pol}, size={self.size}, dtype={self._raw('signal').dtype}, noise={'NULL' if self._raw('noise') is NULL else 'array'}){where}"
