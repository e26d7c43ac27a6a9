"""N > 1 path on CPU: two gloo ranks shard independent propagations round-robin and gather them
in unit order (opticomlib_amd.dist); every rank must end up with every unit's result, equal to a
serial run."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from opticomlib_amd import dist as od
from oracle import ssfm_numpy as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_round_robin():
    assert od.shard(8, 0, 1) == list(range(8))
    assert od.shard(8, 1, 2) == [1, 3, 5, 7]
    assert od.shard(5, 3, 4) == [3]
    assert od.shard(2, 3, 4) == []
    units = sorted(u for r in range(8) for u in od.shard(64, r, 8))
    assert units == list(range(64))
    assert od.world() == (0, 1)
    assert od.sharded_map(lambda u: np.array([u]), 3) == [np.array([0]), np.array([1]), np.array([2])]


@pytest.mark.parametrize("n_units", [5, 2])
def test_two_rank_gloo_matches_serial(tmp_path, n_units):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _dist_worker import field
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_dist_worker.py"), str(tmp_path), str(n_units)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    kw = dict(length=3, h=1.0, alpha=0.2, beta_2=-20.0, gamma=2.0)
    want = [orc.fiber_c64(field(u), 6.25e-12, **kw) for u in range(n_units)]
    for rank in (0, 1):
        got = np.load(os.path.join(tmp_path, f"rank{rank}.npz"))
        assert len(got.files) == n_units
        for u in range(n_units):
            np.testing.assert_array_equal(got[f"arr_{u}"], want[u])
