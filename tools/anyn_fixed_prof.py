"""Where the fixed costs of a FIBER() call on a long length that is not a power of two go (cProfile of 2-step calls: transfers and pinned allocations, 2.3 ms).  GPU box."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, '/root/repo')
import numpy as np
import opticomlib_amd as oa
from opticomlib_amd import workloads
from opticomlib_amd.typing import gv, optical_signal
gv(**workloads.BENCH_GV)
n = ((1 << 16) - 1) * 16
a = workloads.qpsk_field(n, seed=1616, n_pol=2, power_w=2e-3)
x = optical_signal(a)
kw = dict(length=2 * 0.125, h=0.125, **workloads.SMF)
for _ in range(3): oa.FIBER(x, **kw).signal
t = time.perf_counter(); y = oa.FIBER(x, **kw); t1 = time.perf_counter(); s = y.signal; t2 = time.perf_counter()
print(f"2-step call: FIBER() {1e3*(t1-t):.2f} ms, .signal {1e3*(t2-t1):.2f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(5): oa.FIBER(x, **kw).signal
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
