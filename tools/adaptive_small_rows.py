"""Adaptive runs of small / medium plans, us per step and launches (which engine ran): [SSFM_...=..] python tools/adaptive_small_rows.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from opticomlib_amd import _lib, devices, workloads
dt = 1.0 / (16 * 32e9)
for k, rows in ((11, 2), (12, 1), (12, 2), (13, 1), (13, 2), (14, 1), (14, 2)):
    n = 1 << k
    a = workloads.qpsk_field(n, seed=1, n_pol=2, power_w=5e-3)[:rows]
    D = devices.linear_operator(n, dt, 0.2, -21.7, 0.13)
    p = _lib.Plan(n, rows, _lib.C64); p.set_linear_operator(D)
    for rep in range(3):
        p.set_field(a); p.synchronize()
        t = time.perf_counter(); s, z, _ = p.propagate_adaptive(1.3, 40.0, 0.002, False); el = time.perf_counter() - t
    print(f"2^{k} x {rows}: {s} steps, {el / s * 1e6:.2f} us per step, {p.last_propagate_ms()[1]} launches", flush=True)
    p.close()
