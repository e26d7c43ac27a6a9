import sys; sys.path.insert(0, '.')
import numpy as np, warnings
import opticomlib_amd as oa
from opticomlib_amd import workloads
from opticomlib_amd.typing import gv, optical_signal
from oracle import ssfm_numpy as orc
gv(**workloads.BENCH_GV)
a = workloads.qpsk_field(1 << 12, seed=5, power_w=1.0)
for gamma, h, L in ((10.0, 1.0, 2.0), (100.0, 5.0, 5.0), (1000.0, 50.0, 50.0), (2e4, 50.0, 100.0)):
    kw = dict(length=L, h=h, alpha=0.0, beta_2=-21.7, beta_3=0.0, gamma=gamma)
    y = oa.FIBER(optical_signal(a), **kw).signal
    r = orc.fiber_c64(a, gv.dt, **kw)
    phi = gamma * np.max(np.abs(a) ** 2) * h / 2
    print(f"gamma={gamma:g} h={h:g}: max phase per half step {phi:.3g} rad, max|d|/peak = {np.max(np.abs(y - r)) / np.max(np.abs(r)):.2e}, float32 ulp of the phase = {np.spacing(np.float32(phi)):.1e} rad")
