"""Phase timeline of the one-launch filter kernel (dev aid; needs a -DSOS_TIMELINE=1 build of sos_filter.hip):
    VSRC=sos_filter.hip tools/variants.sh build tl:-DSOS_TIMELINE=1;  SSFM_LIB=build/var/_ssfm_tl.so [LOG2N=20 ROWS=2 CPLX=1] python tools/sos_timeline.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from opticomlib_amd import _lib
from scipy import signal as sg
n, cplx, rows = 1 << int(os.environ.get("LOG2N", "20")), os.environ.get("CPLX", "1") == "1", int(os.environ.get("ROWS", "2"))
sos = sg.bessel(4, 0.05, "low", norm="mag", output="sos"); zi = sg.sosfilt_zi(sos)
dt = np.complex128 if cplx else np.float64
x = _lib.DeviceArray.from_host(np.random.default_rng(1).standard_normal((rows, n)).astype(dt), dt, 0)
y = _lib.DeviceArray(x.shape, dt, 0)
for _ in range(20): _lib.sosfiltfilt_device(sos, zi, x.ptr, y.ptr, n, rows, cplx, 0)
os.environ["SOS_TIMELINE_DUMP"] = os.environ.get("SOS_TL_PATH", "1")          # (a path: every workgroup's line goes there)
_lib.sosfiltfilt_device(sos, zi, x.ptr, y.ptr, n, rows, cplx, 0)
print("kernels %.1f us" % (_lib.sosfiltfilt_last_ms() * 1e3))
