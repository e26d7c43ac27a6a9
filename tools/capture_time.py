"""z-resolved capture beside the run (ssfm_propagate_fixed_capture, round 5): configuration C2 (2^20 x 2 complex64, 1000 steps) plain, with a snapshot every
100 / 50 / 10 steps into page-locked memory, with the scalar log, with both -- wall time per SSFM step, best of three; and the every-step capture of the
reference's return_steps on a shorter run for scale.     python tools/capture_time.py  ->  gpurun_out/r05_capture_time.txt"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import opticomlib_amd as oa
from opticomlib_amd import _lib, workloads
from opticomlib_amd.typing import gv

gv(**workloads.BENCH_GV)
n = 1 << 20
a = workloads.qpsk_field(n, seed=2024).astype(np.complex64)
hs = np.full(1000, 0.125, np.float32)
p = _lib.Plan(n, 2, _lib.C64)
p.set_linear_operator(oa.devices.linear_operator(n, gv.dt, **{k: workloads.SMF[k] for k in ("alpha", "beta_2", "beta_3")}))
out = []


def best(f, reps=3):
    t = 1e9
    for _ in range(reps):
        p.set_field(a); p.synchronize()
        t0 = time.perf_counter(); f(); p.synchronize()
        t = min(t, time.perf_counter() - t0)
    return t


t0 = time.perf_counter()
for _ in range(10):
    p.get_field()
d2h = (time.perf_counter() - t0) / 10
out.append(f"for scale: one 16 MiB snapshot from the device into page-locked memory takes {d2h * 1e3:.2f} ms here ({16.78 / d2h / 1e3:.1f} GB/s) = {d2h / 15.4e-6:.0f} steps of the run")
plain = best(lambda: p.propagate_fixed(1.3, hs))
p.set_field(a); p.propagate_fixed(1.3, hs); p.synchronize(); ref = p.get_field().copy()
out.append(f"plain run, 1000 steps: {plain * 1e3:.1f} us per step")
for label, kw in (("snapshot every 100 steps (11 x 16 MiB)", dict(every=100)), ("every 50 (21 x 16 MiB)", dict(every=50)), ("every 10 (101 x 16 MiB)", dict(every=10)),
                  ("scalar log only", dict(scalars=True)), ("every 100 + scalar log", dict(every=100, scalars=True))):
    t = best(lambda: p.propagate_fixed_capture(1.3, hs, **kw))
    same = np.array_equal(p.get_field(), ref)
    out.append(f"{label}: {t * 1e3:.1f} us per step ({(t / plain - 1) * 100:+.1f} %), end field {'bit-identical to' if same else 'DIFFERS from'} the plain run's")
hs100 = hs[:100]
t_all = best(lambda: p.propagate_fixed(1.3, hs100, snapshots=True), reps=2)
t_100 = best(lambda: p.propagate_fixed(1.3, hs100))
out.append(f"for scale, 100 steps: plain {t_100 * 1e4:.1f} us per step; the reference's every-step capture (101 x 16 MiB, three launches per step, pageable destination) {t_all * 1e4:.1f} us per step")
p.close()
print("\n".join(out))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "r05_capture_time.txt"), "w").write("\n".join(out) + "\n")
