"""Why two correct ADAPTIVE runs of a band full of noise end 3e-5 apart (round 6): step sizes of the HIP run, of the float64 solution of the adaptive problem
(fuzz_cases.truth_adaptive_f64) and of the oracle side by side -- they agree to 1e-6 until the LAST, clamped step, whose length follows the float32 rounding of the
accumulated z (devices.py:1196: min(h, L - z)): 3e-6 km of fibre, 8e-5 rad of dispersion at the edge of a 512 GHz band.      python tests/diag/adaptive_long_diag.py"""
import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/tests/diag')
import numpy as np
import opticomlib_amd as oa
from opticomlib_amd import workloads
from opticomlib_amd.typing import gv, optical_signal
from oracle import ssfm_numpy as orc
import fuzz_cases
gv(**workloads.BENCH_GV)
n = 100003
rng = np.random.default_rng(n + 5)
a = ((rng.standard_normal((2, n)) + 1j * rng.standard_normal((2, n))) * 0.05).astype(np.complex64)
kw = dict(length=12, phi_max=0.02, **workloads.SMF)
z, A_z = oa.FIBER(optical_signal(a), return_steps=True, **kw)
y = oa.FIBER(optical_signal(a), **kw).signal
zr, Ar = orc.fiber_c64(a, gv.dt, return_steps=True, **kw)
rel = lambda u, v: np.abs(u - v).max() / np.abs(v).max()
print("steps", len(z) - 1, len(zr) - 1, "max |z - zr| / z", np.max(np.abs(z[1:] - zr[1:]) / zr[1:]), "max |dh|/h", np.max(np.abs(np.diff(z) - np.diff(zr)) / np.diff(zr)))
t_own = fuzz_cases.truth_f64(a, gv.dt, np.diff(np.asarray(z, dtype=np.float64)), kw)
t_orc = fuzz_cases.truth_f64(a, gv.dt, np.diff(np.asarray(zr, dtype=np.float64)), kw)
print("y vs truth(own schedule)", rel(y, t_own), " oracle vs truth(its schedule)", rel(Ar[-1], t_orc), " truth(own) vs truth(oracle's)", rel(t_own, t_orc), " y vs oracle", rel(y, Ar[-1]))
zt, truth = fuzz_cases.truth_adaptive_f64(a, gv.dt, kw)
print("truth steps", len(zt) - 1, "max |z - zt| / z", np.max(np.abs(z[1:] - zt[1:]) / zt[1:]), "max |dh|/h vs truth", np.max(np.abs(np.diff(z) - np.diff(zt)) / np.diff(zt)))
print("y vs adaptive truth", rel(y, truth), " oracle vs adaptive truth", rel(Ar[-1], truth))
# fixed-step control: the same field, the truth's own schedule as a fixed schedule through truth_f64 and through FIBER one step at a time is not possible; compare instead a FIXED h run
kwf = dict(length=12, h=0.3, **workloads.SMF)
yf = oa.FIBER(optical_signal(a), **kwf).signal
from opticomlib_amd import devices, _lib
hs, _ = devices.step_schedule(12, 0.3, _lib.C64)
tf = fuzz_cases.truth_f64(a, gv.dt, hs, kwf)
of = orc.fiber_c64(a, gv.dt, **kwf)
print("fixed h=0.3, 40 steps: y vs truth", rel(yf, tf), " oracle vs truth", rel(of, tf), " y vs oracle", rel(yf, of))
hz, ht, hr = np.diff(z), np.diff(zt), np.diff(zr)
print("step: h ours, h truth, h oracle, (ours-truth)/h, (oracle-truth)/h")
for k in range(0, len(hz), 3):
    print(f"{k:3d} {hz[k]:.7f} {ht[k]:.7f} {hr[k]:.7f} {(hz[k]-ht[k])/ht[k]:+.2e} {(hr[k]-ht[k])/ht[k]:+.2e}")
