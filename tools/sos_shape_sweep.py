"""Zero-phase filter: kernel time over sizes for three workgroup shapes (short chunk x 1 or 4 wavefronts, long chunk x 4) and the dispatcher's own choice (which also has the long chunk x 1), to place the dispatcher's thresholds.
    python tools/sos_shape_sweep.py  (GPU box)"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from opticomlib_amd import _lib
from scipy import signal as sg
ORDER = int(os.environ.get("ORDER", "4"))
sos = sg.bessel(ORDER, 0.05, "low", norm="mag", output="sos"); zi = sg.sosfilt_zi(sos)
variants = (("short x1", {"SOS_WAVES_FORCE": "1"}), ("short x4", {"SOS_WAVES_FORCE": "4", "SSFM_SOS_LONG_CHUNK": "0"}), ("long x4", {"SOS_WAVES_FORCE": "4"}), ("default", {}))
print("shape".ljust(28) + "".join(v[0].rjust(12) for v in variants) + "   (kernels, us; * = three launches)")
for cplx, rows in ((False, 1), (True, 1), (True, 2)):
    for lg in range(int(os.environ.get("LG0", "13")), 21):
        for n in ((1 << lg), 3 << (lg - 1)):
            if n > 1 << 20: continue
            dt = np.complex128 if cplx else np.float64
            x = _lib.DeviceArray.from_host(np.random.default_rng(1).standard_normal((rows, n)).astype(dt), dt, 0)
            y = _lib.DeviceArray(x.shape, dt, 0)
            line = f"n={n:8d} rows={rows} {'c128' if cplx else 'f64 '}".ljust(28)
            for name, env in variants:
                for k in ("SOS_WAVES_FORCE", "SSFM_SOS_LONG_CHUNK"): os.environ.pop(k, None)
                os.environ.update(env)
                best = []
                for rep in range(3):
                    for _ in range(10): _lib.sosfiltfilt_device(sos, zi, x.ptr, y.ptr, n, rows, cplx, 0)
                    t = 0.0
                    for _ in range(40):
                        _lib.sosfiltfilt_device(sos, zi, x.ptr, y.ptr, n, rows, cplx, 0); t += _lib.sosfiltfilt_last_ms()
                    best.append(t / 40 * 1e3)
                line += f"{min(best):11.1f}{'*' if _lib.sosfiltfilt_last_launches() != 1 else ' '}"
            print(line, flush=True)
            x.close() if hasattr(x, "close") else None; y.close() if hasattr(y, "close") else None
