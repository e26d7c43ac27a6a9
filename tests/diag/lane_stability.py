"""The lanes test of the GPU suite, many times in one process: fresh two-lane plans of 2^19 x 2 beside 0 ... 5 other live plans, every plan's
step time and what its lane health check had to do.   python tests/diag/lane_stability.py [rounds]  ->  gpurun_out/lane_stability.txt"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["SSFM_LANES"] = "2"
import opticomlib_amd as oa
from opticomlib_amd import _lib, workloads
from opticomlib_amd.typing import gv

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
gv(**workloads.BENCH_GV)
n = 1 << 19
a = workloads.qpsk_field(n, seed=4).astype(np.complex64)
D = oa.devices.linear_operator(n, gv.dt, 0.2, -21.7, 0.13)
D14 = oa.devices.linear_operator(1 << 14, gv.dt, 0.2, -21.7, 0.13)
hs = np.full(300, 0.125, np.float32)
out = ["# round | other plans alive | us per step: first run, best of the next three | lane_score alone_us pair_us last_us | heals remade dropped shared | plan creation ms"]
slow = remade = heals = dropped = 0
best_all = None
rows = []
for r in range(rounds):
    others = []
    for k in range(6):
        t0 = time.perf_counter()
        p = _lib.Plan(n, 2, _lib.C64)
        t_make = (time.perf_counter() - t0) * 1e3
        p.set_linear_operator(D); p.set_field(a)
        t0 = time.perf_counter(); p.propagate_fixed(1.3, hs); p.synchronize(); first = (time.perf_counter() - t0) / 300 * 1e6
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); p.propagate_fixed(1.3, hs); p.synchronize()
            best = min(best, (time.perf_counter() - t0) / 300 * 1e6)
        i = p.last_run_info()
        p.close()
        rows.append((r, k, first, best, i, t_make))
        best_all = best if best_all is None else min(best_all, best)
        q = _lib.Plan(1 << 14, 1, _lib.C64)
        q.set_linear_operator(D14); q.set_field(workloads.qpsk_field(1 << 14, seed=k, n_pol=1)); q.propagate_fixed(1.3, hs[:3]); q.synchronize()
        others.append(q)
    for q in others:
        q.close()
for r, k, first, best, i, t_make in rows:
    bad = best > 1.35 * best_all
    slow += bad; remade += i["lanes_remade"]; heals += i["lane_heals"]; dropped += i["lanes_dropped"]
    if bad or i["lane_heals"] or r < 2 or r % 25 == 0:
        out.append(f"{r:4d} | {k} | {first:7.2f} {best:7.2f}{' SLOW' if bad else ''} | {i['lane_score']:.2f} {i['lane_alone_us']:.2f} {i['lane_pair_us']:.2f} {i['lane_last_us']:.2f} | "
                   f"{i['lane_heals']} {i['lanes_remade']} {int(i['lanes_dropped'])} {int(i['lanes_share_queue'])} | {t_make:.1f}")
bests = np.array([x[3] for x in rows]); firsts = np.array([x[2] for x in rows]); makes = np.array([x[5] for x in rows])
out.append(f"# {len(rows)} plans in {rounds} rounds: best-of-three us per step min {bests.min():.2f} median {np.median(bests):.2f} max {bests.max():.2f}; first run median {np.median(firsts):.2f} max {firsts.max():.2f}; "
           f"plans still slow after their first run (> 1.35 x the best plan): {slow}; run-time heals {heals}, streams remade at run time {remade}, plans dropped to one lane {dropped}; "
           f"plan creation median {np.median(makes):.1f} ms max {makes.max():.1f} ms; pairs from the pool: {sum(1 for x in rows if x[4].get('lanes_from_pool'))}, "
           f"ratings at plan creation in the process: {rows[-1][4].get('lane_ratings_total')}")
print("\n".join(out[-12:]))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "lane_stability.txt"), "w").write("\n".join(out) + "\n")
