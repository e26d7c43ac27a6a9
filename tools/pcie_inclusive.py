"""FIBER() / DM() end to end from host arrays to host arrays (PCIe-inclusive), C2 and C1."""
import sys, time; sys.path.insert(0, '.')
import numpy as np
import opticomlib_amd as oa
from opticomlib_amd import workloads
from opticomlib_amd.typing import gv, optical_signal
gv(**workloads.BENCH_GV)
a = workloads.qpsk_field(1 << 20, seed=2024, n_pol=2, power_w=1e-3)
for label, kw in (("C2 (complex64, 1000 steps)", dict(length=125, h=0.125)), ("C1 (complex128, 100 steps)", dict(length=100, h=1.0, precision="complex128"))):
    best = 1e9
    for _ in range(4):
        x = optical_signal(a.copy())
        t = time.perf_counter(); y = oa.FIBER(x, **kw, **workloads.SMF).signal; best = min(best, time.perf_counter() - t)
    steps = 1000 if "C2" in label else 100
    print(f"{label}: {best * 1e3:.2f} ms end to end -> {(1 << 20) * steps / best / 1e9:.1f} G sample-steps/s")
best = 1e9
for _ in range(4):
    x = optical_signal(a.copy())
    t = time.perf_counter(); y = oa.DM(x, D=-2000).signal; best = min(best, time.perf_counter() - t)
print(f"DM 2^20 x 2 complex128: {best * 1e3:.2f} ms end to end")
