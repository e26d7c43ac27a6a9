"""Worker of tests/test_dist_cpu.py: world_size-2 gloo run of the sharding harness.  The compute
function is the oracle (tests may use it); the harness under test is opticomlib_amd.dist."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from opticomlib_amd import dist as od  # noqa: E402
from oracle import ssfm_numpy as orc  # noqa: E402


def field(u, n=512):
    rng = np.random.default_rng(3000 + u)
    return ((rng.standard_normal((2, n)) + 1j * rng.standard_normal((2, n))) * 0.03).astype(np.complex64)


def main():
    out_dir = sys.argv[1]
    n_units = int(sys.argv[2])
    rank, ws = od.init("gloo")
    assert ws == 2
    dt = 6.25e-12
    kw = dict(length=3, h=1.0, alpha=0.2, beta_2=-20.0, gamma=2.0)
    mine = od.shard(n_units)
    assert mine == list(range(rank, n_units, ws))
    res = od.sharded_map(lambda u: orc.fiber_c64(field(u), dt, **kw), n_units, to_all=True)
    assert len(res) == n_units
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), *res)
    res0 = od.sharded_map(lambda u: np.full((3,), float(u)), n_units, to_all=False)
    assert (res0 is None) == (rank != 0)
    if rank == 0:
        assert [float(r[0]) for r in res0] == [float(u) for u in range(n_units)]
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
