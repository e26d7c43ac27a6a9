import sys, time; sys.path.insert(0, '.')
import numpy as np
import opticomlib_amd as oa
from opticomlib_amd import workloads
from opticomlib_amd.typing import gv, optical_signal
gv(**workloads.BENCH_GV)
for k, steps in ((10, 1000), (12, 1000), (14, 400), (16, 200), (20, 20)):
    a = workloads.qpsk_field(1 << k, seed=1)
    x = optical_signal(a)
    kw = dict(length=steps * 0.5, h=0.5, **workloads.SMF)
    oa.FIBER(x, return_steps=True, **kw)
    t = time.perf_counter(); z, A = oa.FIBER(x, return_steps=True, **kw); el = time.perf_counter() - t
    y = oa.FIBER(x, **kw).signal
    print(f"2^{k} x 2, {steps} steps with return_steps: {el*1e3:.1f} ms ({el/steps*1e6:.0f} us/step), A_z {A.nbytes/2**20:.0f} MiB, last == plain run within {np.max(np.abs(A[-1]-y))/np.max(np.abs(y)):.1e}")
