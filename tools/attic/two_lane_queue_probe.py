"""Does the opt-in two-lane fused adaptive engine (SSFM_ADAPT_FUSED_LANES=1) depend on which hardware queues its two streams land on?  k other plans
(a stream each, more for two-lane ones) are made first; then the launch count of a 2^19 x 2 adaptive run: 4 per step = two lanes, 3 per step = it fell back (dev aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["SSFM_LANES"] = "2"; os.environ["SSFM_ADAPT_FUSED_LANES"] = "1"
import numpy as np
import opticomlib_amd as oa
from opticomlib_amd import _lib, workloads
from opticomlib_amd.typing import gv
gv(**workloads.BENCH_GV)
n = 1 << 19
a = workloads.qpsk_field(n, seed=109, power_w=10e-3)
D = oa.devices.linear_operator(n, gv.dt, 0.2, -21.7, 0.13)
keep = []
for k in range(9):
    p = _lib.Plan(n, 2, _lib.C64)
    p.set_linear_operator(D)
    for rep in range(2):
        p.set_field(a)
        steps, z, _ = p.propagate_adaptive(1.3, 3.0, 0.004, False)
    print(f"{k} other plans alive: {steps} steps, {p.last_propagate_ms()[1]} launches", flush=True)
    p.close()
    keep.append(_lib.Plan(1 << 12, 1, _lib.C128))
