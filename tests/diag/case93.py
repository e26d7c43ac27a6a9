import os, sys, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import opticomlib_amd as oa
from opticomlib_amd import workloads
from opticomlib_amd.typing import gv, optical_signal
from oracle import ssfm_numpy as orc
gv(**workloads.BENCH_GV)
def rel(a, b): return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
rng = np.random.default_rng(1234)
n = 4096
a = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.01
kw = dict(length=30.975502914958835, alpha=0.19581451810912975, beta_2=-12.738706221730258, beta_3=0.0, gamma=2.5228050478157877, phi_max=0.005)
zr, Ar = orc.fiber_c64(a, gv.dt, return_steps=True, **kw)
z, Az = oa.FIBER(optical_signal(a), return_steps=True, **kw)
print("steps", len(zr) - 1, len(z) - 1)
for k in range(1, min(len(z), len(zr))):
    print(f" step {k}: z {zr[k]:.7f} {z[k]:.7f}  dz_rel {abs(zr[k]-z[k])/zr[k]:.1e}  field diff {rel(Az[k], Ar[k]):.2e}")
# the same physical run with the oracle's own step sizes replayed as a FIXED schedule through the ABI
from opticomlib_amd import _lib, devices
hs = np.diff(zr).astype(np.float32)
p = _lib.Plan(n, 1, _lib.C64); p.set_linear_operator(orc.linear_operator_c64(n, gv.dt, kw["alpha"], kw["beta_2"], kw["beta_3"])); p.set_field(a)
p.propagate_fixed(kw["gamma"], hs); y = p.get_field()[0]; p.close()
print("fixed replay of the oracle's schedule vs oracle:", rel(y, Ar[-1]))
# band-limited input, same fibre
b = workloads.qpsk_field(n, seed=3, n_pol=1, power_w=2e-4)[0]
zr, Ar = orc.fiber_c64(b, gv.dt, return_steps=True, **kw)
y = oa.FIBER(optical_signal(b), **kw).signal
print("band-limited input, adaptive:", len(zr) - 1, "steps,", rel(y, Ar[-1]))
