"""The one-launch filter's hand-over many times over (round 5): its group totals cross between workgroups as 16-byte units {value, tag} with no flag and no
acknowledgement awaited -- a total taken half-written or from an earlier call would show as a wrong result.  Thousands of calls back to back, the shapes
alternating (so that the same buffers serve different grids), every result compared with SciPy's.
     python tests/diag/filter_repeat.py [calls]  ->  gpurun_out/r05_filter_repeat.txt"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from opticomlib_amd import _lib
from scipy import signal as sg

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
rng = np.random.default_rng(3)
shapes = [(1 << 20, True, 2, 4), (1 << 20, False, 1, 4), (1 << 18, True, 2, 2), (1 << 16, False, 1, 4), (1 << 14, True, 1, 4), (300007, True, 1, 3), (1 << 19, False, 3, 4), (70001, False, 2, 1)]
cases = []
for n, cplx, rows, order in shapes:
    sos = sg.bessel(order, 0.04 + 0.02 * len(cases), "low", norm="mag", output="sos"); zi = sg.sosfilt_zi(sos)
    dt = np.complex128 if cplx else np.float64
    xh = (rng.standard_normal((rows, n)) + (1j * rng.standard_normal((rows, n)) if cplx else 0)).astype(dt)
    want = sg.sosfiltfilt(sos, xh, axis=-1)
    x = _lib.DeviceArray.from_host(xh, dt, 0); y = _lib.DeviceArray(x.shape, dt, 0)
    cases.append((n, cplx, rows, sos, zi, x, y, want, np.abs(want).max()))
worst, bad, forms = 0.0, 0, {}
t0 = time.time()
for c in range(calls):
    n, cplx, rows, sos, zi, x, y, want, peak = cases[int(rng.integers(len(cases)))]
    _lib.sosfiltfilt_device(sos, zi, x.ptr, y.ptr, n, rows, cplx, 0)
    forms[_lib.sosfiltfilt_last_launches()] = forms.get(_lib.sosfiltfilt_last_launches(), 0) + 1
    if c % 8 == 0 or c > calls - 64:                       # (a read-back of 32 MiB per call would be the whole run time: every eighth call, and the last 64)
        e = float(np.abs(y.to_host() - want).max() / peak)
        worst = max(worst, e)
        bad += e > 1e-11
out = [f"# {calls} calls of ssfm_sosfiltfilt back to back over {len(shapes)} shapes (2^14 ... 2^20 samples, 1-3 rows, real / complex, orders 1-4) in random order, {time.time() - t0:.0f} s;",
       f"# results read back and compared with scipy.signal.sosfiltfilt on every eighth call and the last 64: worst {worst:.1e} of the peak, beyond 1e-11: {bad}; calls by launches per call: {forms}"]
print("\n".join(out))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "r05_filter_repeat.txt"), "w").write("\n".join(out) + "\n")
sys.exit(1 if bad else 0)
