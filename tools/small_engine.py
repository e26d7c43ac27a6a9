"""The single-launch engine of small plans (k_small) against the two-kernel engine (SSFM_SMALL=0 at plan creation):
us per step of a 1000-step fixed schedule, 2^8 ... 2^13 samples, 1 / 2 / 64 rows, complex64 and complex128."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from opticomlib_amd import _lib, devices
dt = 1.0 / (16 * 32e9)
for prec, name in ((_lib.C64, "c64"), (_lib.C128, "c128")):
    for k in (8, 10, 12, 13):
        n = 1 << k
        for rows in (1, 2, 64):
            rng = np.random.default_rng(k)
            a = ((rng.standard_normal((rows, n)) + 1j * rng.standard_normal((rows, n))) * 0.05)
            D = devices.linear_operator(n, dt, 0.2, -21.7, 0.13, prec)
            hs, _ = devices.step_schedule(1000 * 0.05, 0.05, prec)
            out = []
            for small in ("1", "0"):
                os.environ["SSFM_SMALL"] = small
                p = _lib.Plan(n, rows, prec); p.set_linear_operator(D); p.set_field(a)
                p.propagate_fixed(1.3, hs); p.synchronize()
                t = time.perf_counter(); p.propagate_fixed(1.3, hs); p.synchronize(); el = (time.perf_counter() - t) / hs.size
                out.append((el * 1e6, p.last_propagate_ms()[1]))
                p.close()
            print(f"{name} 2^{k} x {rows}: single launch {out[0][0]:6.2f} us/step ({out[0][1]} launch)   two kernels {out[1][0]:6.2f} us/step ({out[1][1]} launches)", flush=True)
