#!/usr/bin/env python3
"""Full-size fixture of configuration C4: one Monte-Carlo realisation -- the QPSK-like field whose bits
come from the reference's LFSR (PRBS(order=15, seed=5)), 2^20 x 2, complex64 -- through FIBER(100 x 1 km) and then
DBP(100 x 1 km), the reference-semantics back-propagation (NOT the transmitted field: the stale-N^ round trip is not
the identity).  Every 257th sample, per-polarisation power and energy of the FIBER output and of the DBP output.

Produced by the ORACLE (oracle/ssfm_numpy, bit-exact with the imported reference on every captured vector) from the
ORACLE's LFSR (oracle/prbs_numpy, pinned to the reference's own PRBS vectors); the pulse shaping is the host NumPy
code of opticomlib_amd.workloads (input generation, not the path under test).  About 4 minutes on one core.

    python tests/golden/make_c4_strided.py
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from opticomlib_amd import workloads  # noqa: E402
from oracle import prbs_numpy as po  # noqa: E402
from oracle import ssfm_numpy as orc  # noqa: E402

SEED, ORDER, SPS, N_POL = 5, 15, 16, 2
n = 1 << 20
dt = 1.0 / (workloads.BENCH_GV["sps"] * workloads.BENCH_GV["R"])
nsym = n // SPS
bits, _ = po.prbs(ORDER, 2 * N_POL * nsym, SEED)
b = bits.reshape(N_POL, nsym, 2).astype(np.int64)
sym = ((2 * b[..., 0] - 1) + 1j * (2 * b[..., 1] - 1)) / np.sqrt(2)
a = workloads._shape_pulses(sym, n, SPS, 1e-3)
kw = dict(length=100, h=1.0, **workloads.SMF)
t = time.time()
y = orc.fiber_c64(a, dt, **kw)
x = orc.dbp_c64(y, dt, **kw)
print(f"oracle C4 realisation (100 + 100 steps): {time.time() - t:.0f} s")


def summary(o):
    o2 = np.abs(o.astype(np.complex128)) ** 2
    return o[:, ::257], np.mean(o2, axis=-1), np.sum(o2)


ys, yp, ye = summary(y)
xs, xp, xe = summary(x)
np.savez(os.path.join(HERE, "c4_full_strided.npz"), input_samples=a[:, ::257].astype(np.complex64), fiber_samples=ys, fiber_power=yp, fiber_energy=ye,
         dbp_samples=xs, dbp_power=xp, dbp_energy=xe, seed=SEED, _versions=np.array([np.__version__]))
