"""Per-step time of a fixed-step run, straight through the C ABI (no torch): the quick A/B probe.
    [SSFM_LIB=...] [FIELDS=1] [PREC=c64|c128] [STEPS=1000] [LOG2N=20] [POL=2] [REPS=3] python tools/step_time.py [label]
prints `label: X us per field-step` (median and min over REPS runs of the schedule)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from opticomlib_amd import _lib, devices, workloads

n = 1 << int(os.environ.get("LOG2N", "20"))
pol = int(os.environ.get("POL", "2"))
fields = int(os.environ.get("FIELDS", "1"))
steps = int(os.environ.get("STEPS", "1000"))
reps = int(os.environ.get("REPS", "3"))
prec = _lib.C128 if os.environ.get("PREC", "c64") == "c128" else _lib.C64
dt = 1.0 / (16 * 32e9)
hs = np.full(steps, 0.125, np.float32 if prec == _lib.C64 else np.float64)
D = devices.linear_operator(n, dt, 0.2, -21.7, 0.13, prec)
a = np.concatenate([workloads.qpsk_field(n, seed=s, n_pol=pol) for s in range(fields)]).astype(np.complex64 if prec == _lib.C64 else np.complex128)
p = _lib.Plan(n, pol * fields, prec)
p.set_linear_operator(D)
p.set_field(a)
p.propagate_fixed(1.3, hs)
p.synchronize()
ts = []
enq = []
for _ in range(reps):
    t = time.perf_counter()
    p.propagate_fixed(1.3, hs)
    enq.append((time.perf_counter() - t) / steps / fields * 1e6)      # the host's enqueue loop alone (the call is asynchronous)
    p.synchronize()
    ts.append((time.perf_counter() - t) / steps / fields * 1e6)
label = sys.argv[1] if len(sys.argv) > 1 else "run"
print(f"{label}: {np.median(ts):.2f} us per field-step (min {min(ts):.2f}; {fields} field(s) of 2^{int(np.log2(n))} x {pol}, {'c128' if prec == _lib.C128 else 'c64'}, {steps} steps; host enqueue {np.median(enq):.2f} us per step)", flush=True)
