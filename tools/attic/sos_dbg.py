import os, sys; sys.path.insert(0, "/root/repo")
import numpy as np
from opticomlib_amd import _lib
from scipy import signal as sg
os.environ["SSFM_SOS_DEBUG"] = "1"
for n, cplx, rows in ((1 << 16, False, 1), (1 << 14, True, 1), (1 << 20, True, 2)):
    sos = sg.bessel(4, 0.05, "low", norm="mag", output="sos"); zi = sg.sosfilt_zi(sos)
    dt = np.complex128 if cplx else np.float64
    x = _lib.DeviceArray.from_host(np.random.default_rng(1).standard_normal((rows, n)).astype(dt), dt, 0)
    y = _lib.DeviceArray(x.shape, dt, 0)
    for _ in range(3): _lib.sosfiltfilt_device(sos, zi, x.ptr, y.ptr, n, rows, cplx, 0)
    print("kernels %.1f us, launches" % (_lib.sosfiltfilt_last_ms() * 1e3))
