"""Synthetic link inputs of the benchmark configurations.  ``qpsk_field`` / ``prbs_field`` build
them with NumPy on the host (seeded ``default_rng`` symbols cannot be generated anywhere else);
``prbs_field_device`` builds the Monte-Carlo realisations of configuration C4 in GPU memory from nothing but the
LFSR seed, so a realisation costs no host time and no upload."""
from __future__ import annotations

import numpy as np

# fibre of BASELINE.md section 3
SMF = dict(alpha=0.2, beta_2=-21.7, beta_3=0.13, gamma=1.3)
BENCH_GV = dict(sps=16, R=32e9)          # fs = 512 GHz, dt = 1.953125 ps


def _shape_pulses(sym: np.ndarray, n: int, sps: int, power_w: float) -> np.ndarray:
    x = np.zeros(sym.shape[:-1] + (n,), dtype=np.complex128)
    x[..., sps // 2::sps] = sym
    f = np.fft.fftfreq(n) * sps                                    # in symbol rates
    x = np.fft.ifft(np.fft.fft(x, axis=-1) * np.exp(-(f / 0.6) ** 2 * np.log(2)), axis=-1)
    x *= np.sqrt(power_w / np.mean(np.abs(x) ** 2, axis=-1, keepdims=True))
    return x


def qpsk_field(n: int, seed: int, n_pol: int = 2, sps: int = 16, power_w: float = 1e-3) -> np.ndarray:
    """QPSK-like dual-polarisation field: ``n/sps`` random symbols per polarisation, Gaussian
    spectral shaping ``exp(-(f/0.6R)^2 ln2)``, 0 dBm (1 mW) mean power per polarisation."""
    rng = np.random.default_rng(seed)
    b = rng.integers(0, 2, size=(n_pol, n // sps, 2))
    sym = ((2 * b[..., 0] - 1) + 1j * (2 * b[..., 1] - 1)) / np.sqrt(2)
    return _shape_pulses(sym, n, sps, power_w)


def lfsr_bits(order: int, length: int, seed: int) -> np.ndarray:
    """``PRBS(order, length, seed)`` as a uint8 array (the reference's LFSR, ``devices.py:166-175``), for the
    Monte-Carlo realisations of configuration C4."""
    from .devices import PRBS
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", UserWarning)          # seed 0 -> 1
        return PRBS(order, int(length), int(seed)).data


def prbs_field(n: int, seed: int, n_pol: int = 2, sps: int = 16, power_w: float = 1e-3, order: int = 15) -> np.ndarray:
    """QPSK-like field whose bits come from an LFSR seeded per realisation."""
    nsym = n // sps
    bits = lfsr_bits(order, 2 * n_pol * nsym, seed).reshape(n_pol, nsym, 2).astype(np.int64)
    sym = ((2 * bits[..., 0] - 1) + 1j * (2 * bits[..., 1] - 1)) / np.sqrt(2)
    return _shape_pulses(sym, n, sps, power_w)


def prbs_field_device(n: int, seed: int, n_pol: int = 2, sps: int = 16, power_w: float = 1e-3, order: int = 15, device=None):
    """:func:`prbs_field` generated where it is used: the LFSR bits (``ssfm_prbs``), the QPSK-like symbols
    (``ssfm_load_qpsk``), the Gaussian spectral shaping (one resident transfer function on a complex128 plan) and the
    power normalisation all run on the GPU; returns a ``_lib.DeviceArray`` of shape ``(n_pol, n)``, complex64.
    Agrees with the host version to the rounding of the two FFT implementations (1e-15 relative before the cast)."""
    from . import _lib
    from .devices import _tag, default_device, get_plan
    dev = default_device() if device is None else int(device)
    nsym = n // sps
    s0 = int(seed) % (1 << order) or 1                            # PRBS(): seed modulo 2^order, 0 -> 1
    bits, _ = _lib.prbs_device(order, 2 * n_pol * nsym, s0, dev)
    plan = get_plan(n, n_pol, _lib.C128, dev)
    with plan.lock:
        tag = _tag("gauss-shape", n, sps)
        if plan.tag(1) != tag:
            f = np.fft.fftfreq(n) * sps                           # in symbol rates
            plan.transfer_table(np.exp(-(f / 0.6) ** 2 * np.log(2)).astype(np.complex128), 0)
            plan.set_tag(1, tag)
        plan.load_qpsk(bits, nsym, sps)
        plan.apply_table(0)
        plan.synchronize()
        src = plan.field_device_ptr
        pw = _lib.power_device(src, n_pol, n, True, dev)
        out = _lib.DeviceArray((n_pol, n), np.complex128, dev)
        for r in range(n_pol):
            _lib._check(_lib.load().ssfm_device_scale_add(dev, _lib._VP(out.ptr + r * n * 16), _lib._VP(src + r * n * 16),
                                                          float(np.sqrt(power_w / pw[r])), None, 2 * n), "ssfm_device_scale_add")
    return out.astype(np.complex64)
