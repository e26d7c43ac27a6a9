"""How the error of the complex64 engines grows with the number of steps (against the float64 restatement): the one-launch chirp-z engine at n = 8176
and the power-of-two engine at 8192 / 16384 (whose transforms are half as many and half as long)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import opticomlib_amd as oa
from opticomlib_amd import workloads
from opticomlib_amd.typing import optical_signal, gv
from oracle import ssfm_numpy as orc

gv(**workloads.BENCH_GV)
for n in (8176, 8192, 16384, 3000, 4096):
    a = workloads.qpsk_field(1 << 15, seed=5, power_w=4e-3, n_pol=1)[0, :n]
    x = optical_signal(a)
    row = []
    for steps in (10, 30, 100, 300, 1000):
        kw = dict(length=0.5 * steps, h=0.5, **workloads.SMF)
        y = oa.FIBER(x, **kw).signal
        r128 = orc.fiber_c128(a, gv.dt, **kw)
        r64 = orc.fiber_c64(a, gv.dt, **kw)
        rel = lambda p, q: float(np.max(np.abs(p - q)) / np.max(np.abs(q)))
        row.append(f"{steps}: {rel(y, r128):.2e} (oracle c64 {rel(r64, r128):.2e})")
    print(n, " | ".join(row), flush=True)
