"""Python-level cost of LPF() / BPF() calls on short signals against the kernels' time (dev aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import opticomlib_amd as oa
from opticomlib_amd import _lib
from opticomlib_amd.typing import gv, optical_signal, electrical_signal
gv(sps=16, R=32e9)
for k in (12, 14, 16):
    n = 1 << k
    rng = np.random.default_rng(k)
    xe = electrical_signal(rng.standard_normal(n))
    xo = optical_signal(rng.standard_normal((2, n)) + 1j * rng.standard_normal((2, n)))
    for name, f in (("LPF real", lambda: oa.LPF(xe, BW=20e9)), ("BPF 2 x complex", lambda: oa.BPF(xo, BW=60e9))):
        f().signal
        t = time.perf_counter()
        for _ in range(50): y = f().signal
        el = (time.perf_counter() - t) / 50
        print(f"2^{k} {name}: {el * 1e6:.0f} us per call (kernels {_lib.sosfiltfilt_last_ms() * 1e3:.1f} us)")
