"""z-resolved capture beside the run (ssfm_propagate_fixed_capture, round 5): configuration C2 (2^20 x 2 complex64, 1000 steps) plain, with a snapshot every
100 / 50 / 20 steps into page-locked memory, with the scalar log, with both.  Interleaved rounds; per mode the wall clock from the call to the moment the
caller's buffers are valid (ssfm_synchronize) and the run's own device time (the plan's events around the lanes: what the propagation itself takes).  The
destination is allocated once, outside the timed region (the Python wrapper Plan.propagate_fixed_capture allocates per call: page-locking 176 MiB is not
the capture's cost).  For scale: the plain run followed by one transfer of the end field (a capture run delivers it), and the reference's every-step
capture on a shorter run.     python tools/capture_time.py  ->  gpurun_out/r05_capture_time.txt"""
import os, sys, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import opticomlib_amd as oa
from opticomlib_amd import _lib, workloads
from opticomlib_amd.typing import gv

gv(**workloads.BENCH_GV)
n, nsteps = 1 << 20, 1000
a = workloads.qpsk_field(n, seed=2024).astype(np.complex64)
hs = np.full(nsteps, 0.125, np.float32)
p = _lib.Plan(n, 2, _lib.C64)
p.set_linear_operator(oa.devices.linear_operator(n, gv.dt, **{k: workloads.SMF[k] for k in ("alpha", "beta_2", "beta_3")}))
lib = _lib.load()
fields = _lib.host_empty((1 + nsteps // 20, 2, n), np.complex64, limit=8 << 30)          # the largest capture of the list
scal = _lib.host_empty((nsteps + 1, 2, 2), np.float64)
modes = [("plain", 0, False), ("plain + the end field to the host", -1, False), ("scalar log only", 0, True), ("every 100 (11 x 16 MiB)", 100, False),
         ("every 100 + scalar log", 100, True), ("every 50 (21 x 16 MiB)", 50, False), ("every 20 (51 x 16 MiB: PCIe carries 23 per run)", 20, False)]
res = {m[0]: [] for m in modes}
ends = {}
for r in range(6):
    for name, every, log in modes[r % len(modes):] + modes[:r % len(modes)]:          # (rotated: no mode always runs behind the same other one)
        p.set_field(a); p.propagate_fixed(1.3, hs[:300])              # (untimed: every mode starts on a GPU that has just been busy, whatever ran before it)
        p.set_field(a); p.synchronize()
        t0 = time.perf_counter()
        if every <= 0 and not log:
            p.propagate_fixed(1.3, hs)
            if every < 0:
                p.get_field()
        else:
            cap = _lib.Capture(every, fields.ctypes.data if every else None, scal.ctypes.data if log else None)
            _lib._check(lib.ssfm_propagate_fixed_capture(p._h, 1.3, hs.ctypes.data_as(C.c_void_p), nsteps, C.byref(cap)), "ssfm_propagate_fixed_capture")
        p.synchronize()
        wall = time.perf_counter() - t0
        dev = p.last_propagate_ms()[0] * 1e-3
        if r:
            res[name].append((wall, dev))
        else:
            ends[name] = p.get_field().copy()
out = []
t0 = time.perf_counter()
for _ in range(10):
    p.get_field()
d2h = (time.perf_counter() - t0) / 10
out.append(f"C2, {nsteps} steps, interleaved rounds, best of five.  One 16 MiB snapshot over this box's PCIe: {d2h * 1e3:.2f} ms ({16.78 / d2h / 1e3:.1f} GB/s) = {d2h / 15.4e-6:.0f} steps of the run.")
out.append("mode | wall until the buffers are valid, us per step (against plain) | the run's device time, us per step (against plain) | end field against the plain run's")
w0, d0 = (min(x[i] for x in res["plain"]) for i in (0, 1))
for name, every, log in modes:
    w, d = (min(x[i] for x in res[name]) for i in (0, 1))
    same = "bit-identical" if np.array_equal(ends[name], ends["plain"]) else "%.1e of the peak (the log's kernels are another instantiation)" % (np.abs(ends[name] - ends["plain"]).max() / np.abs(ends["plain"]).max())
    out.append(f"{name} | {w * 1e6 / nsteps:.2f} ({(w / w0 - 1) * 100:+.1f} %) | {d * 1e6 / nsteps:.2f} ({(d / d0 - 1) * 100:+.1f} %) | {same}")
hs100 = hs[:100]
ts = []
for _ in range(2):
    p.set_field(a); p.synchronize(); t0 = time.perf_counter(); p.propagate_fixed(1.3, hs100, snapshots=True); p.synchronize(); ts.append(time.perf_counter() - t0)
out.append(f"for scale, 100 steps: the reference's every-step capture (101 x 16 MiB, three launches per step, pageable destination) {min(ts) * 1e4:.0f} us per step")
p.close()
print("\n".join(out))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "r05_capture_time.txt"), "w").write("\n".join(out) + "\n")
