"""Small plans, fixed step: eager launches against hipGraph replay (run with SSFM_GRAPH=0 / 1); the graph is captured on the
second run of a schedule, so the timing starts at the fourth."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from opticomlib_amd import _lib, devices, workloads
dt = 1.0 / (16 * 32e9)
for k in (10, 12, 14, 16, 18):
    n = 1 << k
    a = workloads.qpsk_field(n, seed=1, n_pol=2, power_w=5e-3)
    p = _lib.Plan(n, 2, _lib.C64); p.set_linear_operator(devices.linear_operator(n, dt, 0.2, -21.7, 0.13)); p.set_field(a)
    hs, _ = devices.step_schedule(500 * 0.1, 0.1)
    for _ in range(3):
        p.propagate_fixed(1.3, hs); p.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        p.propagate_fixed(1.3, hs)
    p.synchronize()
    el = (time.perf_counter() - t) / 3 / hs.size
    print(f"2^{k} x 2, SSFM_GRAPH={os.environ.get('SSFM_GRAPH', '0')}: {el * 1e6:.2f} us/step", flush=True)
    p.close()
