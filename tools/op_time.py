import sys, time; sys.path.insert(0, '.')
import numpy as np
from opticomlib_amd import devices as d, _lib
for prec, name in ((_lib.C64, "c64"), (_lib.C128, "c128")):
    for n in (1 << 16, 1 << 20):
        d._GRID_POWERS.clear()
        t = time.perf_counter(); op = d._linear_operator(n, 1e-11, 0.2, -21.7, 0.1, prec); el0 = time.perf_counter() - t
        t = time.perf_counter(); op = d._linear_operator(n, 1e-11, 0.2, -21.7, 0.1, prec); el = time.perf_counter() - t
        w = np.asarray(np.fft.fftfreq(n, 1e-11) * 2 * np.pi * 1e-12, dtype=np.float32 if prec == _lib.C64 else np.float64)
        t = time.perf_counter(); w3 = w ** 3; e3 = time.perf_counter() - t
        t = time.perf_counter(); w2 = w ** 2; e2 = time.perf_counter() - t
        t = time.perf_counter(); ff = np.fft.fftfreq(n, 1e-11); ef = time.perf_counter() - t
        print(f"{name} n=2^{n.bit_length()-1}: operator {el0*1e3:.2f} ms on a new grid, {el*1e3:.2f} ms after (w**3 {e3*1e3:.2f}, w**2 {e2*1e3:.2f}, fftfreq {ef*1e3:.2f})")
import opticomlib_amd as oa
from opticomlib_amd import workloads
from opticomlib_amd.typing import gv, optical_signal
gv(**workloads.BENCH_GV)
a = workloads.qpsk_field(1 << 20, seed=1, n_pol=2, power_w=1e-3)
x = optical_signal(a)
oa.FIBER(x, length=1, h=1.0, **workloads.SMF)
for k in range(3):
    d._OPERATORS.clear()                                   # a new fibre on the same grid (w**2, w**3 are kept per grid)
    for pl in d._PLANS.values(): pl._op_key = None
    t = time.perf_counter(); y = oa.FIBER(x, length=100, h=1.0, **workloads.SMF); e1 = time.perf_counter() - t
    t = time.perf_counter(); y = oa.FIBER(x, length=100, h=1.0, **workloads.SMF); e2 = time.perf_counter() - t
    print(f"FIBER 2^20 x 2, 100 steps from a host array: first call of a fibre {e1*1e3:.2f} ms, again {e2*1e3:.2f} ms")
