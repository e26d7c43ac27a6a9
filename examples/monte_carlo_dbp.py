"""Monte-Carlo realisations of one link, sharded over the GPUs of a node (configuration C4 of BASELINE.json):

    for every PRBS seed:  QPSK-like field -> FIBER (fixed step) -> DBP with the same fibre     [no host round trip in between]

Independent realisations never exchange data: realisation i runs on rank i % world, one gather at the end.

    python examples/monte_carlo_dbp.py [realisations] [log2 samples]                       # one GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/monte_carlo_dbp.py 64 20
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticomlib_amd import dist, workloads  # noqa: E402

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rank, world = dist.init()
dt = 1.0 / (workloads.BENCH_GV["sps"] * workloads.BENCH_GV["R"])
from opticomlib_amd import _lib  # noqa: E402
n = 1 << log2n
mine = dist.shard(seeds)                                                                             # this rank's realisations
# every realisation is generated in GPU memory from its LFSR seed (PRBS bits, symbols, pulse shaping: HIP kernels)
local = _lib.DeviceArray((len(mine), 2, n), np.complex64)
for k, u in enumerate(mine):
    f = workloads.prbs_field_device(n, seed=u + 1)
    _lib._check(_lib.load().ssfm_device_copy(local.device, _lib._VP(local.ptr + k * f.nbytes), _lib._VP(f.ptr), f.nbytes, 2), "ssfm_device_copy")
link = dict(length=100.0, h=1.0, **workloads.SMF)

t0 = time.perf_counter()
received = dist.propagate_channels(local, dt, dbp=False, to_all=False, **link)                      # what the receiver sees (gathered on rank 0)
restored = dist.propagate_channels(local, dt, dbp=True, to_all=False, **link)                       # ... after digital back-propagation
elapsed = time.perf_counter() - t0
if rank == 0:
    fields = [workloads.prbs_field_device(n, seed=u + 1).to_host() for u in range(seeds)]          # the transmitted fields, for the comparison
    peak = np.max(np.abs(fields))
    err_rx = max(np.max(np.abs(r - f)) for r, f in zip(received, fields)) / peak
    err_dbp = max(np.max(np.abs(r - f)) for r, f in zip(restored, fields)) / peak
    print(f"{seeds} realisations of 2^{log2n} x 2 samples on {world} GPU(s): {elapsed * 1e3:.1f} ms for FIBER and FIBER+DBP")
    print(f"distance from the transmitted field: {err_rx:.2f} of its peak after the fibre, {err_dbp:.1e} after back-propagation")
import torch.distributed as td  # noqa: E402
if td.is_available() and td.is_initialized():
    td.destroy_process_group()
