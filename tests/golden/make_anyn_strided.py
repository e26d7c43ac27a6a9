#!/usr/bin/env python3
"""Full-size fixtures of a length that is NOT a power of two (round 6): the reference's PRBS-16 word at 16 samples per bit, (2^16 - 1) * 16 = 1 048 560 samples per
polarisation, dual-pol -- every 4099th sample and the per-row power of the ORACLE's run (oracle/ssfm_numpy.fiber_c64, bit-exact with the imported reference on every
captured vector) of
  * FIBER(length=12.5, h=0.125): 100 of configuration C2's steps,
  * FIBER(length=10.0, phi_max=0.002): the adaptive run (z log, samples, power).
The input is regenerated from its seed (workloads.qpsk_field), so the file holds outputs only.

    python tests/golden/make_anyn_strided.py
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
STRIDE = 4099
N = ((1 << 16) - 1) * 16
out = {}


def pw(y):
    return np.mean(np.abs(y.astype(np.complex128)) ** 2, axis=-1)


a = workloads.qpsk_field(N, seed=1616, n_pol=2, power_w=2e-3)
t = time.time()
y = orc.fiber_c64(a, dt, length=12.5, h=0.125, **workloads.SMF)
out["fixed_samples"], out["fixed_power"] = y[:, ::STRIDE], pw(y)
print(f"{N} x 2, 100 fixed steps: {time.time() - t:.0f} s", flush=True)
t = time.time()
z, A_z = orc.fiber_c64(a, dt, length=10.0, phi_max=0.002, return_steps=True, **workloads.SMF)
out["adapt_z"], out["adapt_samples"], out["adapt_power"] = np.asarray(z), A_z[-1][:, ::STRIDE], pw(A_z[-1])
print(f"{N} x 2, adaptive, {len(z) - 1} steps: {time.time() - t:.0f} s", flush=True)
np.savez(os.path.join(HERE, "anyn_strided.npz"), _versions=np.array([np.__version__]), stride=np.array(STRIDE), n=np.array(N), **out)
