"""Timings of the other BASELINE configurations on one GPU (dev aid): C1 (complex128, 100 steps),
adaptive mode, DM, and the host-inclusive FIBER() call."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from opticomlib_amd import _lib, devices, workloads, FIBER, DBP, DM, gv, optical_signal
n = 1 << 20
gv(**workloads.BENCH_GV)
a = workloads.qpsk_field(n, seed=2024)
D64 = devices.linear_operator(n, gv.dt, 0.2, -21.7, 0.13, _lib.C64)
D128 = devices.linear_operator(n, gv.dt, 0.2, -21.7, 0.13, _lib.C128)

def timeit(f, reps=3):
    f(); t = time.perf_counter()
    for _ in range(reps): f()
    return (time.perf_counter() - t) / reps

# C1: complex128, 100 x 1 km
p = _lib.Plan(n, 2, _lib.C128); p.set_linear_operator(D128); p.set_field(a)
hs, _ = devices.step_schedule(100, 1.0, _lib.C128)
def run():
    p.propagate_fixed(1.3, hs); p.synchronize()
t = timeit(run); print(f"C1 c128 100 steps: {t*1e3:.2f} ms -> {t/100*1e6:.1f} us/step, {n*100/t/1e9:.1f} G sample*steps/s")
p.close()
# C2-like adaptive (10 mW to force many steps)
p = _lib.Plan(n, 2, _lib.C64); p.set_linear_operator(D64)
b = (a * np.sqrt(10)).astype(np.complex64)
def run():
    p.set_field(b); return p.propagate_adaptive(1.3, 20.0, 0.01, False)
t = timeit(run, 2); s, z, _ = run(); print(f"adaptive c64 L=20 phi=0.01: {s} steps in {t*1e3:.2f} ms -> {t/s*1e6:.1f} us/step")
p.close()
x = optical_signal(a)
t = timeit(lambda: FIBER(x, length=125, h=0.125, **workloads.SMF)); print(f"FIBER() host-inclusive C2: {t*1e3:.1f} ms -> {n*1000/t/1e9:.1f} G sample*steps/s")
t = timeit(lambda: DM(x, D=-21.7*80)); print(f"DM() host-inclusive 2^20 x 2 c128: {t*1e3:.1f} ms")
p = _lib.Plan(n, 2, _lib.C128); p.set_field(a); H = np.ones(n, complex)
t = timeit(lambda: p.apply_transfer(H)); print(f"apply_transfer (H upload + 3 kernels): {t*1e3:.2f} ms")
from opticomlib_amd import LPF, BPF
from scipy import signal as sg
t = timeit(lambda: BPF(x, BW=60e9)); sos = sg.bessel(4, 30e9, 'low', fs=gv.fs, norm='mag', output='sos')
t0 = time.perf_counter(); sg.sosfiltfilt(sos, a, axis=-1); tc = time.perf_counter() - t0
print(f"BPF() 2^20 x 2 c128 host-inclusive: {t*1e3:.1f} ms (scipy on this host: {tc*1e3:.0f} ms)")
pw = np.abs(a[0]) ** 2
t = timeit(lambda: LPF(pw, BW=20e9)); t0 = time.perf_counter(); sg.sosfiltfilt(sos, pw); tc = time.perf_counter() - t0
print(f"LPF() 2^20 real host-inclusive: {t*1e3:.1f} ms (scipy: {tc*1e3:.0f} ms)")
from opticomlib_amd import PD, EDFA
xn = optical_signal(a, 0.1 * a[::-1].copy())
np.random.seed(0); t = timeit(lambda: PD(xn, BW=20e9, include_noise="none")); print(f"PD(include_noise='none') 2^20 x 2 host-inclusive: {t*1e3:.1f} ms")
t = timeit(lambda: PD(xn, BW=20e9)); print(f"PD('all': 2 x 2^20 host normal variates) host-inclusive: {t*1e3:.1f} ms")
t = timeit(lambda: EDFA(xn, G=20, NF=5, BW=100e9)); print(f"EDFA(G, NF, BW) 2^20 x 2 host-inclusive (4 x 2^20 host normal variates): {t*1e3:.1f} ms")
# a short link as a script writes it: host array in, detector voltage read at the end
from opticomlib_amd import devices as od
def link():
    y = FIBER(x, length=100, h=1.0, **workloads.SMF)
    z = DBP(y, length=100, h=1.0, **workloads.SMF)
    f = BPF(z, BW=100e9)
    return PD(f, BW=20e9, include_noise="none").signal
for keep in (False, True):
    od.KEEP_ON_DEVICE = keep
    t = timeit(link); print(f"FIBER(100) -> DBP(100) -> BPF -> PD, 2^20 x 2, results {'kept on the device' if keep else 'returned to the host after every call'}: {t*1e3:.1f} ms")
od.KEEP_ON_DEVICE = True
t = timeit(lambda: PD(xn, BW=20e9, rng="device").signal); print(f"PD('all', rng='device') 2^20 x 2 host-inclusive, voltage read: {t*1e3:.1f} ms")
t = timeit(lambda: EDFA(xn, G=20, NF=5, BW=100e9, rng="device")); print(f"EDFA(G, NF, BW, rng='device') 2^20 x 2 (result left on the device): {t*1e3:.1f} ms")
