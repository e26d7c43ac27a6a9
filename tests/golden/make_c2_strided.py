#!/usr/bin/env python3
"""Full-size fixture of configuration C2 (2^20 x 2 complex64, 1000 steps): every 257th sample, the per-polarisation
power and sum |out|^2, as SURVEY.md 8(c)(2) asks.  Produced by the ORACLE (oracle/ssfm_numpy.fiber_c64, itself
bit-exact with the imported reference on every captured vector) -- a run of the reference itself at this size takes
just as long and gives the same bits.  About 15 minutes on one core.

    python tests/golden/make_c2_strided.py
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from opticomlib_amd import workloads  # noqa: E402
from oracle import ssfm_numpy as orc  # noqa: E402

n = 1 << 20
dt = 1.0 / (workloads.BENCH_GV["sps"] * workloads.BENCH_GV["R"])
a = workloads.qpsk_field(n, seed=2024)
t = time.time()
out = orc.fiber_c64(a, dt, length=125, h=0.125, **workloads.SMF)
print(f"oracle C2: {time.time() - t:.0f} s")
np.savez(os.path.join(HERE, "c2_full_strided.npz"), samples=out[:, ::257], power=np.mean(np.abs(out.astype(np.complex128)) ** 2, axis=-1),
         energy=np.sum(np.abs(out.astype(np.complex128)) ** 2), _versions=np.array([np.__version__]))

# configuration C1 (the same field, complex128, 100 x 1 km) from the float64 restatement
t = time.time()
out1 = orc.fiber_c128(a, dt, length=100, h=1.0, **workloads.SMF)
print(f"oracle C1: {time.time() - t:.0f} s")
np.savez(os.path.join(HERE, "c1_full_strided.npz"), samples=out1[:, ::257], power=np.mean(np.abs(out1) ** 2, axis=-1), energy=np.sum(np.abs(out1) ** 2),
         _versions=np.array([np.__version__]))
