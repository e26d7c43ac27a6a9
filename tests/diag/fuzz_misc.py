"""One-off stress of the remaining paths: complex128 propagation, large lengths that are not powers of two, DAC
pulse shaping -- each against its oracle (the device-resident chains are covered by tests/test_gpu_parity.py)."""
import os, sys, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import opticomlib_amd as oa
from opticomlib_amd import workloads
from opticomlib_amd.typing import gv, optical_signal
from oracle import ssfm_numpy as orc, transmitter_numpy as tx

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
gv(**workloads.BENCH_GV)
def rel(a, b): return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
bad = 0
# complex128, fixed steps, any length
w = []
for i in range(40):
    n = int(rng.choice([1 << int(rng.integers(8, 16)), rng.integers(2, 30000)]))
    npol = int(rng.integers(1, 3))
    a = workloads.qpsk_field(1 << max(6, (n - 1).bit_length()), seed=int(rng.integers(1 << 30)), n_pol=npol, power_w=float(rng.choice([1e-4, 1e-3, 1e-2])))[:, :n]
    a = a[0] if npol == 1 else a
    kw = dict(length=float(rng.uniform(1, 20)), h=float(rng.choice([0.3, 1.0, 2.5])), alpha=float(rng.uniform(0, 0.4)), beta_2=float(rng.uniform(-25, 25)),
              beta_3=float(rng.uniform(-0.3, 0.3)), gamma=float(rng.uniform(0, 3)))
    y = oa.FIBER(optical_signal(a), precision="complex128", **kw).signal
    w.append((rel(y, orc.fiber_c128(a, gv.dt, **kw)), n, npol))
w.sort(key=lambda t: -t[0]); print("complex128 worst:", ["%.1e n=%d x %d" % t for t in w[:3]]); bad += sum(t[0] > 1e-10 for t in w)
# large lengths that are not powers of two (few steps: the oracle is slow there)
w = []
for n in (100003, 262145, 500000, 1000003):
    a = workloads.qpsk_field(1 << (n - 1).bit_length(), seed=n, n_pol=1, power_w=2e-3)[0, :n]
    kw = dict(length=3.0, h=1.0, **workloads.SMF)
    w.append((rel(oa.FIBER(optical_signal(a), **kw).signal, orc.fiber_c64(a, gv.dt, **kw)), n))
print("large odd lengths:", ["%.1e n=%d" % t for t in w]); bad += sum(t[0] > 2e-5 for t in w)
# DAC
gv(sps=int(rng.choice([8, 16, 32])), R=10e9)
w = []
for i in range(40):
    bits = oa.PRBS(int(rng.choice([7, 9, 15])), len=int(rng.integers(3, 4000)), seed=int(rng.integers(1, 1 << 20))).data
    shape = str(rng.choice(["nrz", "gaussian", "rcos"]))
    kw = dict(pulse_shape=shape, Vpp=float(rng.uniform(0.1, 10)), offset=float(rng.uniform(-5, 5)), coupling=str(rng.choice(["AC", "DC"])))
    if shape == "nrz":
        kw["T"] = int(rng.integers(1, 3))
    if shape == "gaussian":
        kw.update(T=int(rng.integers(1, 3)), m=int(rng.integers(1, 3)), c=float(rng.choice([0.0, 0.5])) if kw.get("m", 1) == 1 else 0.0)
        if kw["m"] > 1:
            kw["c"] = 0.0
    if shape == "rcos":
        kw.update(beta=float(rng.uniform(0.05, 1.0)), rcos_type=str(rng.choice(["normal", "sqrt"])))
    got = oa.DAC(bits, **kw).signal
    want = tx.dac(bits, gv.sps, gv.fs, **kw)
    w.append((rel(got, want), bits.size, shape))
w.sort(key=lambda t: -t[0]); print("DAC worst:", ["%.1e bits=%d %s" % t for t in w[:3]]); bad += sum(t[0] > 1e-11 for t in w)
print("violations:", bad)
sys.exit(1 if bad else 0)
