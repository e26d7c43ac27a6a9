#!/usr/bin/env python3
"""Full-size fixtures of ADAPTIVE runs (the reference's default mode, h=None): a 2^20 x 2 complex64 field, FIBER(length=6, phi_max=0.004), and
the 2^19 x 1 / 2^18 x 2 grids of the fused column kernel's other sizes -- the z log, every 257th sample, per-polarisation power and energy of the
ORACLE's run (oracle/ssfm_numpy.fiber_c64, bit-exact with the imported reference on every captured vector).  Inputs as
tests/test_gpu_parity.py::test_fused_adaptive_column_kernel_on_large_grids makes them.  About a minute on one core.

    python tests/golden/make_adaptive_strided.py
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from opticomlib_amd import workloads  # noqa: E402
from oracle import ssfm_numpy as orc  # noqa: E402

dt = 1.0 / (workloads.BENCH_GV["sps"] * workloads.BENCH_GV["R"])
out = {}
for log2n, rows in ((18, 2), (19, 1), (19, 2), (20, 2)):
    n = 1 << log2n
    a = workloads.qpsk_field(n, seed=70 + log2n, power_w=10e-3)[:rows]
    t = time.time()
    z, Az = orc.fiber_c64(a if rows > 1 else a[0], dt, length=6.0, phi_max=0.004, return_steps=True, **workloads.SMF)
    y = np.asarray(Az[-1]).reshape(rows, n)
    print(f"oracle adaptive 2^{log2n} x {rows}: {len(z) - 1} steps, {time.time() - t:.0f} s")
    k = f"{log2n}x{rows}"
    out[f"z_{k}"] = np.asarray(z, np.float64)
    out[f"samples_{k}"] = y[:, ::257]
    out[f"power_{k}"] = np.mean(np.abs(y.astype(np.complex128)) ** 2, axis=-1)
    out[f"energy_{k}"] = np.sum(np.abs(y.astype(np.complex128)) ** 2)
np.savez(os.path.join(HERE, "adaptive_full_strided.npz"), _versions=np.array([np.__version__]), **out)
