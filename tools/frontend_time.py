"""PD / EDFA at 2^20 x 2, both generators: result left in HBM, and read back to the host."""
import sys, time; sys.path.insert(0, '.')
import numpy as np
import opticomlib_amd as oa
from opticomlib_amd import workloads
from opticomlib_amd.typing import gv, optical_signal
gv(**workloads.BENCH_GV)
a = workloads.qpsk_field(1 << 20, seed=1, n_pol=2, power_w=1e-3)
x = optical_signal(a, 1e-3 * a[::-1])

def timed(label, f, reps=3):
    f().signal
    t = time.perf_counter()
    for _ in range(reps):
        y = f()                                             # every library call is synchronous: the result is complete in HBM
    dev = (time.perf_counter() - t) / reps
    t = time.perf_counter()
    for _ in range(reps):
        y = f(); y.signal; y.noise
    print(f"{label:28s} {dev * 1e3:8.2f} ms on the device, {(time.perf_counter() - t) / reps * 1e3:8.2f} ms with signal and noise read back")

for rng in ("numpy", "device"):
    timed(f"PD('all', rng={rng})", lambda: oa.PD(x, BW=20e9, rng=rng))
    timed(f"EDFA(BW, rng={rng})", lambda: oa.EDFA(x, G=20, NF=5, BW=100e9, rng=rng))
timed("PD('none')", lambda: oa.PD(x, BW=20e9, include_noise="none"))
