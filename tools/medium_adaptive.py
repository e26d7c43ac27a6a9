"""Adaptive runs of plans of 2^14 ... 2^19 samples: the fused column kernel (TM_MID_A, two launches per step) against the
three-launch engine (SSFM_ADAPT_FUSED=0 at plan creation), us per step of a run of a few hundred steps."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from opticomlib_amd import _lib, devices, workloads
dt = 1.0 / (16 * 32e9)
for prec, name in ((_lib.C64, "c64"), (_lib.C128, "c128")):
    for k, rows in ((14, 1), (14, 2), (15, 1), (16, 1), (16, 2), (17, 2), (18, 2), (19, 1)):
        n = 1 << k
        a = workloads.qpsk_field(n, seed=1, n_pol=2, power_w=5e-3)[:rows]
        D = devices.linear_operator(n, dt, 0.2, -21.7, 0.13, prec)
        out = []
        for fused in ("1", "0"):
            os.environ["SSFM_ADAPT_FUSED"] = fused
            p = _lib.Plan(n, rows, prec); p.set_linear_operator(D)
            for rep in range(3):
                p.set_field(a); p.synchronize()
                t = time.perf_counter(); s, z, _ = p.propagate_adaptive(1.3, 40.0, 0.002, False); el = time.perf_counter() - t
            out.append((el / s * 1e6, s, p.last_propagate_ms()[1]))
            p.close()
        print(f"{name} 2^{k} x {rows}: fused {out[0][0]:6.2f} us/step ({out[0][1]} steps, {out[0][2]} launches)   three launches {out[1][0]:6.2f} us/step ({out[1][1]} steps, {out[1][2]} launches)", flush=True)
