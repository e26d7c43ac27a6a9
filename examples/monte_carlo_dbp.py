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
fields = np.stack([workloads.prbs_field(1 << log2n, seed=s + 1) for s in range(seeds)])          # (seeds, 2, N): the same on every rank
link = dict(length=100.0, h=1.0, **workloads.SMF)

t0 = time.perf_counter()
received = dist.propagate_channels(fields, dt, dbp=False, **link)                                   # what the receiver sees
restored = dist.propagate_channels(fields, dt, dbp=True, **link)                                    # ... after digital back-propagation
elapsed = time.perf_counter() - t0
if rank == 0:
    peak = np.max(np.abs(fields))
    err_rx = max(np.max(np.abs(r - f)) for r, f in zip(received, fields)) / peak
    err_dbp = max(np.max(np.abs(r - f)) for r, f in zip(restored, fields)) / peak
    print(f"{seeds} realisations of 2^{log2n} x 2 samples on {world} GPU(s): {elapsed * 1e3:.1f} ms for FIBER and FIBER+DBP")
    print(f"distance from the transmitted field: {err_rx:.2f} of its peak after the fibre, {err_dbp:.1e} after back-propagation")
if world > 1:
    import torch.distributed as td
    td.destroy_process_group()
