"""Zero-phase filter kernels for a handful of shapes, device-resident, back to back (tuning aid for SOS_CHUNK / SOS_WAVES)."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from opticomlib_amd import _lib
from scipy import signal as sg
for n, cplx, rows in ((1 << 20, False, 1), (1 << 20, True, 2), (1 << 18, True, 2), (1 << 16, False, 1), (1 << 14, True, 1)):
    sos = sg.bessel(4, 0.05, "low", norm="mag", output="sos"); zi = sg.sosfilt_zi(sos)
    dt = np.complex128 if cplx else np.float64
    x = _lib.DeviceArray.from_host(np.random.default_rng(1).standard_normal((rows, n)).astype(dt), dt, 0)
    y = _lib.DeviceArray(x.shape, dt, 0)
    for _ in range(20): _lib.sosfiltfilt_device(sos, zi, x.ptr, y.ptr, n, rows, cplx, 0)
    t = time.perf_counter()
    for _ in range(200): _lib.sosfiltfilt_device(sos, zi, x.ptr, y.ptr, n, rows, cplx, 0)
    print(f"n=2^{n.bit_length()-1} rows={rows} complex={cplx}: {(time.perf_counter()-t)/200*1e6:7.1f} us per call, kernels {_lib.sosfiltfilt_last_ms()*1e3:6.1f} us")
