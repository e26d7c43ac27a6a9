"""Is the single-field run bound by the host's launch rate?  Time at which propagate_fixed (asynchronous) returns against the
time at which the device has finished, C2 (2^20 x 2 complex64, 1000 steps = 4002 launches), and the bare launch rate of the
same call sequence on a tiny plan (2^10: the kernels take ~1 us, the host is the limit there)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from opticomlib_amd import _lib, devices, workloads
dt = 1.0 / (16 * 32e9)
for k, lanes in ((20, None), (10, "2"), (10, "1")):
    if lanes: os.environ["SSFM_LANES"] = lanes
    n = 1 << k
    a = workloads.qpsk_field(n, seed=2024) if k >= 14 else (np.random.default_rng(1).standard_normal((2, n)) * 0.03).astype(np.complex64)
    p = _lib.Plan(n, 2, _lib.C64); p.set_linear_operator(devices.linear_operator(n, dt, 0.2, -21.7, 0.13)); p.set_field(a.astype(np.complex64))
    hs, _ = devices.step_schedule(125, 0.125)
    for _ in range(2): p.propagate_fixed(1.3, hs); p.synchronize()
    for rep in range(3):
        t0 = time.perf_counter(); p.propagate_fixed(1.3, hs); t1 = time.perf_counter(); p.synchronize(); t2 = time.perf_counter()
        ms, launches = p.last_propagate_ms()
        print(f"2^{k} x 2 lanes={p.lanes}: propagate_fixed returned after {1e3*(t1-t0):.2f} ms ({1e6*(t1-t0)/launches:.2f} us per launch, {launches} launches), device done after {1e3*(t2-t0):.2f} ms (device time {ms:.2f} ms)")
    p.close()
