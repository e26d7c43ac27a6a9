"""Worker of the multi-rank GPU test: `python -m torch.distributed.run --nproc-per-node W tests/_dist_gpu_worker.py OUT N_UNITS`.
Every rank propagates its share of the units on ITS GPU through opticomlib_amd.dist.propagate_channels (fixed step:
batched in one plan; adaptive: one by one; from host arrays and from device-resident units) and the results are gathered
in GPU memory over RCCL; each rank saves what it received."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from opticomlib_amd import _lib, dist as od, workloads  # noqa: E402
from opticomlib_amd.typing import gv  # noqa: E402


def fields_for(n_units, n=1 << 13):
    return np.stack([workloads.qpsk_field(n, seed=3000 + u) for u in range(n_units)]).astype(np.complex64)


FIXED = dict(length=6, h=0.5, **workloads.SMF)
ADAPT = dict(length=2, alpha=0.2, beta_2=-21.7, gamma=1.3, phi_max=0.005)
DT = 1.0 / (workloads.BENCH_GV["sps"] * workloads.BENCH_GV["R"])


def main():
    out_dir, n_units = sys.argv[1], int(sys.argv[2])
    rank, ws = od.init("nccl")
    gv(sps=4, R=1e9)                                               # a grid of the caller's own: must survive the calls
    saved = (gv.dt, gv.fs)
    f = fields_for(n_units)
    res = {}
    for k, a in enumerate(od.propagate_channels(f, DT, to_all=True, **FIXED)):
        res[f"fixed_{k}"] = a
    got = od.propagate_channels(f, DT, to_all=False, dbp=True, **FIXED)
    assert (got is None) == (rank != 0)
    if rank == 0:
        for k, a in enumerate(got):
            res[f"dbp_{k}"] = a
    for k, a in enumerate(od.propagate_channels(f[:3], DT, to_all=True, **ADAPT)):      # adaptive: one by one, uneven counts
        res[f"adapt_{k}"] = a
    assert (gv.dt, gv.fs) == saved, "propagate_channels changed the caller's sampling grid"
    # units that are already in GPU memory on their rank (generated there): no host array anywhere before the final read
    mine = od.shard(n_units)
    local = _lib.DeviceArray.from_host(f[mine], np.complex64, od_device())
    dev = od.propagate_channels(local, DT, to_all=True, on_device=True, **FIXED)
    assert isinstance(dev, _lib.DeviceArray) and dev.shape == (n_units,) + f.shape[1:]
    res["fixed_dev"] = dev.to_host()
    # the gather itself on known data: every unit carries its own index; uneven counts (padding + reorder), one unit per
    # rank (RCCL writes straight into the result), to every rank and to rank 0 only
    for total in (n_units, ws):
        mine_t = od.shard(total)
        blk = np.stack([np.full((2, 64), u + 1j * (u + 0.5), np.complex64) for u in mine_t]) if mine_t else np.zeros((0, 2, 64), np.complex64)
        d = _lib.DeviceArray.from_host(blk, np.complex64, od_device()) if len(mine_t) else None
        for to_all in (True, False):
            g = od.gather_device(d.ptr if d is not None else 0, len(mine_t), (2, 64), np.complex64, total, od_device(), to_all=to_all, owner=d)
            assert (g is None) == (not to_all and rank != 0)
            if g is not None:
                h = g.to_host()
                assert h.shape == (total, 2, 64)
                for u in range(total):
                    assert np.all(h[u] == np.complex64(u + 1j * (u + 0.5))), (total, to_all, u)
    res["collectives"] = np.array([od.COLLECTIVES["all_gather_into_tensor"], od.COLLECTIVES["gather"]])
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    import torch.distributed as dist
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def od_device():
    from opticomlib_amd.devices import default_device
    return default_device()


if __name__ == "__main__":
    main()
