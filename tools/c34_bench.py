"""Configurations C3 and C4 of BASELINE.json on the GPUs this process group has (dev aid; `bench.py` stays
the C2 contract).  One process: all units on one GPU.  Under
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/c34_bench.py
the units are sharded over N ranks and gathered (opticomlib_amd.dist).

C3: 8 independent 2^20 x 2 complex64 fields (WDM channels, seeds 3000..3007), FIBER(length=125, h=0.125).
C4: 64 PRBS realisations (LFSR seeds 1..64), FIBER(100 x 1 km) then DBP(100 x 1 km), on the device back to back.
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from opticomlib_amd import dist as od, workloads
from opticomlib_amd.typing import gv

rank, ws = od.init()
gv(**workloads.BENCH_GV)
n = 1 << 20

def timed(label, fields, units_steps, **kw):
    od.propagate_channels(fields[: max(ws, 1)], gv.dt, to_all=False, **kw)            # warm-up: plans, tables
    t = time.perf_counter()
    out = od.propagate_channels(fields, gv.dt, to_all=False, **kw)
    el = time.perf_counter() - t
    if rank == 0:
        print(f"{label}: {len(fields)} units on {ws} GPU(s): {el * 1e3:.1f} ms end to end (host arrays in, gathered host arrays out) "
              f"-> {units_steps * n / el / 1e9:.1f} G sample*steps/s", flush=True)
    return out

c3 = np.stack([workloads.qpsk_field(n, seed=3000 + c) for c in range(8)]).astype(np.complex64)
timed("C3", c3, 8 * 1000, length=125, h=0.125, **workloads.SMF)
nseeds = int(os.environ.get("C4_SEEDS", "64"))
per = 8 * ws                                                            # 8 realisations resident per GPU at a time
tot = 0.0
for s0 in range(1, nseeds + 1, per):
    c4 = np.stack([workloads.prbs_field(n, seed=s, power_w=1e-3) for s in range(s0, min(s0 + per, nseeds + 1))]).astype(np.complex64)
    t = time.perf_counter()
    od.propagate_channels(c4, gv.dt, to_all=False, dbp=True, length=100, h=1.0, **workloads.SMF)
    tot += time.perf_counter() - t
if rank == 0:
    print(f"C4: {nseeds} realisations x (100 + 100 steps) on {ws} GPU(s): {tot * 1e3:.1f} ms in propagate_channels "
          f"-> {nseeds * 200 * n / tot / 1e9:.1f} G sample*steps/s (input generation on the host not counted)", flush=True)
