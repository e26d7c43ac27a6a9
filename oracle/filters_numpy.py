"""NumPy restatement of the zero-phase Bessel filters behind ``LPF`` / ``BPF``.  TEST INFRASTRUCTURE.

Reference call sites: ``opticomlib/devices.py:1363-1371`` (LPF) and ``:814-823`` (BPF):
``sg.bessel(N=n, Wn=BW | BW/2, 'low', fs, output='sos', norm='mag')`` then ``sg.sosfiltfilt`` on
``.signal`` and on ``.noise`` separately; LPF keeps ``.real`` and accepts 1-D input only.

The arithmetic lives in a third-party dependency that is not under ``/root/reference``:
**SciPy** (pinned ``scipy==1.12.0`` in the reference's ``requirements.txt:1``; 1.15.3 installed
here).  Restated below from SciPy's published algorithm (``scipy/signal/_signaltools.py``
``sosfiltfilt`` / ``_validate_pad`` / ``odd_ext`` and the direct-form-II-transposed loop of
``_sosfilt.pyx``), with the same operation order, so results are bit-identical to SciPy's C loop on
builds without FMA contraction.  The filter DESIGN (``bessel`` -> SOS, ``sosfilt_zi``) is O(order) work
and is taken from SciPy itself.

Parity status: PINNED by ``tests/golden/lpf_*.npz`` / ``bpf_*.npz`` captured from an import of the
reference (``tests/test_oracle_golden.py``).
"""
from __future__ import annotations

import numpy as np


def bessel_sos(n, cutoff_hz, fs):
    """Design used by both devices (``devices.py:814-816``, ``:1363``); returns ``(sos, zi)``."""
    from scipy import signal as sg
    sos = sg.bessel(N=n, Wn=cutoff_hz, btype="low", fs=fs, output="sos", norm="mag")
    return sos, sg.sosfilt_zi(sos)


def odd_ext(x, edge):
    """``scipy.signal._arraytools.odd_ext`` along the last axis."""
    left = x[..., edge:0:-1]
    right = x[..., -2:-(edge + 2):-1]
    return np.concatenate((2 * x[..., :1] - left, x, 2 * x[..., -1:] - right), axis=-1)


def sosfilt(sos, x, zi):
    """Cascade of biquads, direct form II transposed, sample by sample (``_sosfilt.pyx``).
    ``x``: (..., M); ``zi``: (n_sections, ..., 2).  Vectorised over the leading axes only."""
    x = np.array(x, dtype=np.result_type(x, np.float64))
    z = np.array(zi, dtype=x.dtype)
    y = np.empty_like(x)
    ns = sos.shape[0]
    for n in range(x.shape[-1]):
        xc = x[..., n]
        for s in range(ns):
            xn = xc
            xc = sos[s, 0] * xn + z[s, ..., 0]
            z[s, ..., 0] = sos[s, 1] * xn - sos[s, 4] * xc + z[s, ..., 1]
            z[s, ..., 1] = sos[s, 2] * xn - sos[s, 5] * xc
        y[..., n] = xc
    return y


def sosfiltfilt(sos, zi, x):
    """Forward-backward filtering with odd padding of ``3 * ntaps`` samples and steady-state initial
    conditions (``_signaltools.sosfiltfilt``)."""
    x = np.asarray(x)
    ns = sos.shape[0]
    ntaps = 2 * ns + 1
    ntaps -= min((sos[:, 2] == 0).sum(), (sos[:, 5] == 0).sum())
    edge = ntaps * 3
    if x.shape[-1] <= edge:
        raise ValueError("The length of the input vector x must be greater than padlen, which is %d." % edge)
    ext = odd_ext(x, edge)
    z = zi.reshape((ns,) + (1,) * (x.ndim - 1) + (2,))
    y = sosfilt(sos, ext, z * ext[..., :1].reshape((1,) + ext.shape[:-1] + (1,)))
    y0 = y[..., -1:]
    y = sosfilt(sos, y[..., ::-1], z * y0.reshape((1,) + y.shape[:-1] + (1,)))
    y = y[..., ::-1]
    return y[..., edge:-edge]


def lpf(signal, bw_hz, fs, n=4, noise=None):
    """``LPF`` (``devices.py:1355-1368``): 1-D, real part kept.  Returns ``(signal, noise|None)``."""
    signal = np.asarray(signal)
    if signal.ndim != 1:
        raise ValueError("`input` must be a 1D-array.")
    sos, zi = bessel_sos(n, bw_hz, fs)
    out = sosfiltfilt(sos, zi, signal).real
    out_n = None if noise is None else sosfiltfilt(sos, zi, np.asarray(noise)).real
    return out, out_n


def bpf(signal, bw_hz, fs, n=4, noise=None):
    """``BPF`` (``devices.py:814-823``): cutoff ``BW/2``, complex, last axis."""
    sos, zi = bessel_sos(n, bw_hz / 2, fs)
    out = sosfiltfilt(sos, zi, np.asarray(signal))
    out_n = None if noise is None else sosfiltfilt(sos, zi, np.asarray(noise))
    return out, out_n
