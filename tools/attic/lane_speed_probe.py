"""A fresh two-lane plan of 2^19 x 2 beside K other live one-stream plans (K = 0 ... 9): time per step of a 300-step run, best of three, and what the plan
reports about its lanes' hardware queues.   python tools/attic/lane_speed_probe.py [sub]   (sub: first run another HIP process, as the test before
test_the_lanes_of_a_plan_get_hardware_queues_of_their_own does)"""
import os, sys, time, subprocess
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import opticomlib_amd as oa
from opticomlib_amd import _lib, gv, workloads

gv(**workloads.BENCH_GV)
n = 1 << 19
a = workloads.qpsk_field(n, seed=4).astype(np.complex64)
D = oa.devices.linear_operator(n, gv.dt, 0.2, -21.7, 0.13)
hs = np.full(300, 0.125, np.float32)
D14 = oa.devices.linear_operator(1 << 14, gv.dt, 0.2, -21.7, 0.13)
keep = []
for K in range(10):
    os.environ["SSFM_LANES"] = "2"
    p = _lib.Plan(n, 2, _lib.C64)
    p.set_linear_operator(D); p.set_field(a)
    p.propagate_fixed(1.3, hs); p.synchronize()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter(); p.propagate_fixed(1.3, hs); p.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"K={K}: {min(ts) / 300 * 1e6:6.2f} us per step (worst {max(ts) / 300 * 1e6:6.2f}), lanes_share_queue={p.last_run_info()['lanes_share_queue']}", flush=True)
    p.close()
    os.environ["SSFM_LANES"] = "1"
    q = _lib.Plan(1 << 14, 1, _lib.C64)
    q.set_linear_operator(D14); q.set_field(workloads.qpsk_field(1 << 14, seed=K, n_pol=1)); q.propagate_fixed(1.3, hs[:3]); q.synchronize()
    keep.append(q)
