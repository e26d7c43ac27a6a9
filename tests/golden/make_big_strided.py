#!/usr/bin/env python3
"""Full-size fixtures of the split plans (rows of 2^23 and 2^24 samples; round 6): every 4099th sample and the per-row power of the ORACLE's run
(oracle/ssfm_numpy.fiber_c64, bit-exact with the imported reference on every captured vector) of
  * a 2^24-sample dual-polarisation field, FIBER(length=1.0, h=0.25) = 4 fixed steps (about 15 s per step on one core),
  * the same field, adaptive (h=None, phi_max = 0.002) over 0.5 km -- z log, samples, power,
  * a 2^23-sample single-polarisation field, 4 fixed steps and DBP of the result (reference semantics: not the identity).
Inputs are regenerated from their seeds (workloads.qpsk_field), so the file holds outputs only.

    python tests/golden/make_big_strided.py
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
out = {}


def pw(y):
    return np.mean(np.abs(y.astype(np.complex128)) ** 2, axis=-1)


t = time.time()
a = workloads.qpsk_field(1 << 24, seed=2424, n_pol=2, power_w=4e-3)
kw = dict(length=1.0, h=0.25, **workloads.SMF)
y = orc.fiber_c64(a, dt, **kw)
out["fixed24_samples"], out["fixed24_power"] = y[:, ::STRIDE], pw(y)
print(f"2^24 x 2, 4 fixed steps: {time.time() - t:.0f} s", flush=True)
t = time.time()
z, A_z = orc.fiber_c64(a, dt, length=0.5, phi_max=0.002, return_steps=True, **workloads.SMF)
out["adapt24_z"], out["adapt24_samples"], out["adapt24_power"] = np.asarray(z), A_z[-1][:, ::STRIDE], pw(A_z[-1])
print(f"2^24 x 2, adaptive, {len(z) - 1} steps: {time.time() - t:.0f} s", flush=True)
del A_z, y
t = time.time()
b = workloads.qpsk_field(1 << 23, seed=2323, n_pol=1, power_w=6e-3)[0]
kw = dict(length=2.0, h=0.5, **workloads.SMF)
y = orc.fiber_c64(b, dt, **kw)
x = orc.dbp_c64(y, dt, **kw)
out["fixed23_samples"], out["fixed23_power"] = y[::STRIDE], pw(y)
out["dbp23_samples"], out["dbp23_power"] = x[::STRIDE], pw(x)
print(f"2^23 x 1, 4 fixed steps + DBP: {time.time() - t:.0f} s", flush=True)
np.savez(os.path.join(HERE, "big_strided.npz"), _versions=np.array([np.__version__]), stride=np.array(STRIDE), **out)

# (appended in the same round) the long run: 100 of C2's steps on the 2^23-sample field -- the oracle needs about 3 s per step there
if len(sys.argv) > 1 and sys.argv[1] == "long":
    t = time.time()
    kw = dict(length=12.5, h=0.125, **workloads.SMF)
    y = orc.fiber_c64(b, dt, **kw)
    np.savez(os.path.join(HERE, "big_strided_100.npz"), _versions=np.array([np.__version__]), stride=np.array(STRIDE), samples=y[::STRIDE], power=pw(y))
    print(f"2^23 x 1, 100 fixed steps: {time.time() - t:.0f} s", flush=True)
