"""Throughput vs number of independent fields resident on one GPU (dev aid)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from opticomlib_amd import _lib, devices, workloads
n = 1 << 20
dt = 1.0 / (16 * 32e9)
hs, _ = devices.step_schedule(125.0, 0.125)
D = devices.linear_operator(n, dt, 0.2, -21.7, 0.13)
for fields, lanes in ((1, 1), (1, 2), (2, 2), (2, 4), (4, 2), (4, 4), (4, 8), (8, 4), (8, 8)):
    os.environ["SSFM_LANES"] = str(lanes)
    a = np.concatenate([workloads.qpsk_field(n, seed=s) for s in range(fields)]).astype(np.complex64)
    p = _lib.Plan(n, 2 * fields, _lib.C64)
    p.set_linear_operator(D)
    p.set_field(a)
    p.propagate_fixed(1.3, hs); p.synchronize()
    t = time.perf_counter()
    reps = 3
    for _ in range(reps):
        p.propagate_fixed(1.3, hs)
    p.synchronize()
    el = (time.perf_counter() - t) / reps
    print(f"fields={fields} lanes={lanes}: {el*1e3:.2f} ms per 1000 steps -> {el/1000/fields*1e6:.2f} us per field-step, {fields*n*1000/el/1e9:.1f} G sample*steps/s", flush=True)
    p.close()
