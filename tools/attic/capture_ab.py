"""Where does the time of a capture run go?  Interleaved rounds of plain / scalar log / snapshots at C2, wall clock and the run's own device events
(ssfm_last_propagate_ms: the plan's stream from the first launch to the lanes' join -- the copy stream's tail is not in it).  Round 5 ran it on a
diagnostic build with parts of the capture switched off one at a time (SSFM_X_CAP bits: 1 no transfers to the host, 2 no device copies, 4 no reduce of the
log, 8 no transfer of the last snapshot, 16 the log's launches without the log, 32 no END launches at capture steps): profiles/r05_capture_ab.txt.
     python tools/attic/capture_ab.py <label> [rounds]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
bits = sys.argv[1] if len(sys.argv) > 1 else "0"
os.environ["SSFM_X_CAP"] = bits
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
import opticomlib_amd as oa
from opticomlib_amd import _lib, workloads
from opticomlib_amd.typing import gv

gv(**workloads.BENCH_GV)
n = 1 << 20
a = workloads.qpsk_field(n, seed=2024).astype(np.complex64)
hs = np.full(1000, 0.125, np.float32)
p = _lib.Plan(n, 2, _lib.C64)
p.set_linear_operator(oa.devices.linear_operator(n, gv.dt, **{k: workloads.SMF[k] for k in ("alpha", "beta_2", "beta_3")}))
modes = [("plain", None), ("scal", dict(scalars=True)), ("every1000", dict(every=1000)), ("every100", dict(every=100)), ("every100+scal", dict(every=100, scalars=True))]
res = {m: [] for m, _ in modes}
for r in range(rounds + 1):
    for m, kw in modes:
        p.set_field(a); p.synchronize()
        t0 = time.perf_counter()
        if kw is None:
            p.propagate_fixed(1.3, hs)
        else:
            p.propagate_fixed_capture(1.3, hs, **kw)
        t1 = time.perf_counter()
        p.synchronize()
        t2 = time.perf_counter()
        ms, _ = p.last_propagate_ms()
        if r:
            res[m].append((t2 - t0, t1 - t0, ms * 1e-3))
print(f"SSFM_X_CAP={bits}: ms per 1000-step run: wall (min / median), call returned after (min), device events (min / median)")
for m, _ in modes:
    w = np.array(res[m]) * 1e3
    print(f"  {m:14s} wall {w[:,0].min():6.2f} / {np.median(w[:,0]):6.2f}   returned {w[:,1].min():6.2f}   events {w[:,2].min():6.2f} / {np.median(w[:,2]):6.2f}")
p.close()
