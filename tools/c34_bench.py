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
per = 8                                                                 # 8 realisations resident per GPU at a time
from opticomlib_amd import _lib
mine = od.shard(nseeds)                                                 # this rank's realisations (seeds 1 + unit index)
tot = gen = 0.0
h2d0 = _lib.TRANSFERS["h2d"]
workloads.prbs_field_device(n, seed=1)                                  # (the shaping filter becomes resident: once per process)
h2d0 = _lib.TRANSFERS["h2d"]
for b0 in range(0, len(mine), per):
    chunk = mine[b0:b0 + per]
    t = time.perf_counter()
    block = _lib.DeviceArray((len(chunk), 2, n), np.complex64)
    for k, u in enumerate(chunk):                                       # generated in GPU memory from the LFSR seed: no host array, no upload
        f = workloads.prbs_field_device(n, seed=1 + u, power_w=1e-3)
        _lib._check(_lib.load().ssfm_device_copy(block.device, _lib._VP(block.ptr + k * f.nbytes), _lib._VP(f.ptr), f.nbytes, 2), "ssfm_device_copy")
    gen += time.perf_counter() - t
    t = time.perf_counter()
    out = od.propagate_channels(block, gv.dt, to_all=False, dbp=True, on_device=True, length=100, h=1.0, **workloads.SMF)
    tot += time.perf_counter() - t
if rank == 0:
    print(f"C4: {nseeds} realisations x (100 + 100 steps) on {ws} GPU(s): {tot * 1e3:.1f} ms in propagate_channels (results gathered on the device) "
          f"-> {nseeds * 200 * n / tot / 1e9:.1f} G sample*steps/s; generating the realisations on the device took {gen * 1e3:.1f} ms on top "
          f"({nseeds * 200 * n / (tot + gen) / 1e9:.1f} G with it); host-to-device transfers during the loop: {_lib.TRANSFERS['h2d'] - h2d0}", flush=True)
