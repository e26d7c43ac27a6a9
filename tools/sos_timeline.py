"""Phase timeline of the one-launch filter kernel (dev aid; needs a -DSOS_TIMELINE=1 build of sos_filter.hip):
    VSRC=sos_filter.hip tools/variants.sh build tl:-DSOS_TIMELINE=1;  SSFM_LIB=build/var/_ssfm_tl.so python tools/sos_timeline.py"""
import os, sys; sys.path.insert(0, '.')
import numpy as np
from opticomlib_amd import _lib
from scipy import signal as sg
n, cplx, rows = 1 << 20, True, 2
sos = sg.bessel(4, 0.05, "low", norm="mag", output="sos"); zi = sg.sosfilt_zi(sos)
x = _lib.DeviceArray.from_host(np.random.default_rng(1).standard_normal((rows, n)).astype(np.complex128), np.complex128, 0)
y = _lib.DeviceArray(x.shape, np.complex128, 0)
for _ in range(20): _lib.sosfiltfilt_device(sos, zi, x.ptr, y.ptr, n, rows, cplx, 0)
os.environ["SOS_TIMELINE_DUMP"] = "1"
_lib.sosfiltfilt_device(sos, zi, x.ptr, y.ptr, n, rows, cplx, 0)
print("kernels %.1f us" % (_lib.sosfiltfilt_last_ms() * 1e3))
