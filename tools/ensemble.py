"""Ensembles of short fields on the single-launch engine: rows x 2^k samples, 1000 fixed steps; pol-sample*steps per second."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from opticomlib_amd import _lib, devices
dt = 1.0 / (16 * 32e9)
for k, rows in ((10, 256), (10, 1024), (10, 4096), (12, 256), (12, 512), (12, 1024), (12, 2048), (13, 256), (13, 512)):
    n = 1 << k
    rng = np.random.default_rng(k)
    a = ((rng.standard_normal((rows, n)) + 1j * rng.standard_normal((rows, n))) * 0.05).astype(np.complex64)
    p = _lib.Plan(n, rows, _lib.C64); p.set_linear_operator(devices.linear_operator(n, dt, 0.2, -21.7, 0.13)); p.set_field(a)
    hs, _ = devices.step_schedule(1000 * 0.05, 0.05)
    p.propagate_fixed(1.3, hs); p.synchronize()
    t = time.perf_counter(); p.propagate_fixed(1.3, hs); p.synchronize(); el = (time.perf_counter() - t) / hs.size
    print(f"{rows} x 2^{k}: {el*1e6:7.2f} us/step -> {rows * n / el / 1e9:7.1f} G pol-sample*steps/s ({p.last_propagate_ms()[1]} launch)", flush=True)
    p.close()
