"""One fixed-step (1000 steps) and one adaptive run of the 8176-sample dual-pol field on the one-launch complex64 chirp-z line, for
rocprofv3 --kernel-trace --stats (profiles/r04_chirp_medium_kernel_stats.csv): the run is ONE k_medium_chirp launch between two pointwise ones."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import opticomlib_amd as oa
from opticomlib_amd import workloads
from opticomlib_amd.typing import optical_signal, gv
gv(**workloads.BENCH_GV)
n = 8176
a = workloads.qpsk_field(1 << 14, seed=2, n_pol=2, power_w=4e-3)[:, :n]
x = optical_signal(a)
for kw in (dict(length=500.0, h=0.5, **workloads.SMF), dict(length=20.0, phi_max=0.002, **workloads.SMF)):
    oa.FIBER(x, **kw)
    t = time.perf_counter(); y = oa.FIBER(x, **kw); el = time.perf_counter() - t
    print(f"{kw.get('h', 'adaptive')}: {el * 1e3:.2f} ms, engine {y.engine if hasattr(y, 'engine') else '?'}")
