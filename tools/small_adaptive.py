"""Adaptive runs of small plans: the single-launch kernel (k_small_adapt) against the chunked three-kernel engine
(SSFM_SMALL=0 at plan creation), us per step."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from opticomlib_amd import _lib, devices, workloads
dt = 1.0 / (16 * 32e9)
for prec, name in ((_lib.C64, "c64"), (_lib.C128, "c128")):
    for k, rows in ((8, 2), (10, 1), (10, 2), (12, 1), (12, 2), (13, 1)):
        n = 1 << k
        a = workloads.qpsk_field(n, seed=1, n_pol=2, power_w=5e-3)[:rows]
        D = devices.linear_operator(n, dt, 0.2, -21.7, 0.13, prec)
        out = []
        for small in ("1", "0"):
            os.environ["SSFM_SMALL"] = small
            p = _lib.Plan(n, rows, prec); p.set_linear_operator(D)
            for rep in range(3):
                p.set_field(a); p.synchronize()
                t = time.perf_counter(); s, z, _ = p.propagate_adaptive(1.3, 40.0, 0.002, False); el = time.perf_counter() - t
            out.append((el / s * 1e6, s, p.last_propagate_ms()[1]))
            p.close()
        print(f"{name} 2^{k} x {rows}: single launch {out[0][0]:6.2f} us/step ({out[0][1]} steps, {out[0][2]} launches)   chunked {out[1][0]:6.2f} us/step ({out[1][1]} steps, {out[1][2]} launches)", flush=True)
