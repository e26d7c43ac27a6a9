"""NumPy restatement of the reference's DAC (pulse shaping).  TEST INFRASTRUCTURE.

Reference: ``opticomlib/devices.py:185-350`` (``DAC``), ``opticomlib/utils.py:1791-1947`` (``rcos_pulse``,
``gauss_pulse``, ``nrz_pulse``), ``:1949-1981`` (``upfir``: zero-stuffing at offset ``up//2`` and
``scipy.signal.fftconvolve(xu, h, mode='same')``).  The convolution itself lives in SciPy (pinned 1.12.0 in the
reference's ``requirements.txt``; 1.15.3 installed here) and is taken from there.  Only ``tests/`` may import
this module.

``laser`` / ``mzm`` restate ``devices.py:353-510`` (``LASER``) and ``:620-786`` (``MZM``) with the signal / noise
algebra of ``typing.py:1308-1344`` written out: ``np.cos`` of a signal object acts on ``signal + noise``
(``typing.py:1238-1275``), so the modulator's transfer is formed from the noisy drive voltage and multiplies the
optical signal and the optical noise alike.

Parity status: PINNED by ``tests/golden/dac_*.npz``, ``laser_*.npz``, ``mzm_*.npz`` captured from an import of the reference
(``tests/test_oracle_golden.py``), bit-exact under the same NumPy / SciPy.
"""
from __future__ import annotations

import numpy as np

from . import filters_numpy as fo


def nrz_pulse(span, sps, T):
    N = span * sps
    t = np.linspace(-span / 2, span / 2, N + 1)
    return np.where((t >= -T / 2) & (t < T / 2), 1.0, 0.0)


def gauss_pulse(span, sps, T=1, m=1, c=0.0):
    N = span * sps
    t = np.linspace(-span / 2, span / 2, N + 1)
    alpha = 2 * np.sqrt(np.log(2)) / T
    return np.exp(-(alpha * (1 + 1j * c) * t) ** (2 * m))


def rcos_pulse(beta, span, sps, shape="sqrt"):
    if not (0 <= beta <= 1):
        raise ValueError("beta must be in [0, 1]")
    if shape not in ("sqrt", "normal"):
        raise ValueError("shape must be 'sqrt' or 'normal'")
    N = span * sps
    t = np.linspace(-span / 2, span / 2, N + 1)
    if beta == 0:
        return np.sinc(t)
    if shape == "normal":
        sinc_t = np.sinc(t)
        cos_term = np.cos(np.pi * beta * t)
        den = 1 - (2 * beta * t) ** 2
        p = np.divide(sinc_t * cos_term, den, out=np.zeros_like(den), where=den != 0)
        special = np.abs(den) < 1e-8
        if np.any(special):
            p[special] = (np.pi / 4) * np.sinc(1 / (2 * beta))
        return p
    t_abs = np.abs(t)
    p = np.zeros_like(t)
    mask_zero = t_abs < 1e-8
    p[mask_zero] = (1 - beta) + 4 * beta / np.pi
    mask_special = np.abs(t_abs - 1 / (4 * beta)) < 1e-8
    if np.any(mask_special):
        p[mask_special] = (beta / np.sqrt(2)) * ((1 + 2 / np.pi) * np.sin(np.pi / (4 * beta)) + (1 - 2 / np.pi) * np.cos(np.pi / (4 * beta)))
    general = ~mask_zero & ~mask_special
    if np.any(general):
        ti = t[general]
        num = np.sin(np.pi * ti * (1 - beta)) + 4 * beta * ti * np.cos(np.pi * ti * (1 + beta))
        den = np.pi * ti * (1 - (4 * beta * ti) ** 2)
        p[general] = num / den
    return p


def upfir(x, h, up=1):
    from scipy import signal as sg
    xu = np.zeros(len(x) * up)
    xu[up // 2::up] = x
    return sg.fftconvolve(xu, h, mode="same")


def dac(bits, sps, fs, pulse_shape="nrz", coupling="DC", Vpp=1.0, offset=0.0, h=None, BW=None, **kw):
    """``DAC`` on a 0/1 array.  Returns the output samples (real, or complex for a chirped Gaussian)."""
    bits = np.asarray(bits)
    nbits = bits.size
    span = max(4, nbits - 4)
    if h is not None:
        x = upfir(bits, h=h, up=sps)
    elif pulse_shape.lower() == "nrz":
        x = upfir(bits, h=nrz_pulse(span=span, sps=sps, T=kw.get("T", 1)), up=sps)
    elif pulse_shape.lower() == "gaussian":
        x = upfir(bits, h=gauss_pulse(span=span, sps=sps, T=kw.get("T", 1), m=kw.get("m", 1), c=kw.get("c", 0.0)), up=sps)
    elif pulse_shape.lower() == "rcos":
        x = upfir(bits, h=rcos_pulse(beta=kw.get("beta", 0.25), span=span, sps=sps, shape=kw.get("rcos_type", "normal")), up=sps)
    else:
        raise ValueError(pulse_shape)
    if Vpp is not None:
        x = x * Vpp
    if offset is not None:
        x = x + offset
    if coupling.upper() == "AC":
        x = x - np.mean(x)
    if BW is not None:
        x, _ = fo.lpf(x, BW, fs)
    return x


def time_vector(N, sps, fs):
    """``gv.t`` after ``gv(...)`` (``typing.py:357``)."""
    return np.linspace(0, N * sps / fs, N * sps, endpoint=True)


def laser(t, dt, fs, P0, lw=None, rin=None, df=None):
    """``LASER`` (``devices.py:480-510``); random terms from the global ``np.random`` in the reference's order."""
    out = np.ones_like(t) * np.sqrt(10 ** (P0 / 10 - 3))
    if lw is not None:
        out = out * np.exp(1j * np.cumsum(np.random.normal(0, np.sqrt(2 * np.pi * lw * dt), t.size)))
    if rin is not None:
        rin_noise = np.random.normal(0, np.sqrt(10 ** (rin / 10) * fs), t.size)
        if rin_noise.min() < -1:
            raise ValueError("Noise power is to high, try decrease RIN parameter.")
        out = out * np.sqrt(1 + rin_noise)
    if df is not None:
        if np.abs(df) > fs / 2:
            raise ValueError("The laser frequency is out of the Nyquist range. Try increase the sampling frequency.")
        out = out * np.exp(1j * 2 * np.pi * df * t)
    return out


def mzm(op_signal, op_noise, el_signal, el_noise, fs, bias=0.0, Vpi=5.0, loss_dB=0.0, ER_dB=26.0, pol="x", BW=None):
    """``MZM`` (``devices.py:747-786``).  Returns ``(signal, noise | None)``."""
    loss = 10 ** (-loss_dB / 10)
    eta = 2 * (10 ** (-ER_dB / 10)) ** 0.5
    k = np.pi / 2 / Vpi
    g = k * (np.asarray(el_signal) + bias)
    if el_noise is not None:
        g = g + k * np.asarray(el_noise)
    h = loss ** 0.5 * (np.cos(g) + 1j * eta / 2 * np.sin(g))
    sig = np.asarray(op_signal) * h
    noi = None if op_noise is None else np.asarray(op_noise) * h
    if sig.ndim == 2:
        dead = 1 if pol == "x" else 0
        sig[dead] = np.zeros_like(sig[dead])
        if noi is not None:
            noi[dead] = np.zeros_like(noi[dead])
    if BW is not None:
        sig, noi = fo.bpf(sig, BW, fs, noise=noi)
    return sig, noi
