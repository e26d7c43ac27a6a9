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


@pytest.mark.parametrize("ws", [1, 2, 3, 8])
@pytest.mark.parametrize("n_units", [5, 8, 64, 1])
def test_gather_device_arithmetic_against_a_numpy_model(ws, n_units):
    """The padding and the rank-major -> unit-order copy of dist.gather_device (the one piece of multi-rank logic that a one-GPU box
    never executes with more than one rank), run on CPU tensors for every rank of a simulated world against a plain NumPy model."""
    import torch
    per_unit = 6
    per_rank, unit_order = od.gather_layout(n_units, ws)
    assert per_rank == -(-n_units // ws) and per_rank * ws >= n_units
    units = np.arange(n_units * per_unit, dtype=np.float32).reshape(n_units, per_unit) + 0.5          # unit u = row u
    chunks = []
    for r in range(ws):
        mine = od.shard(n_units, r, ws)
        local = torch.from_numpy(units[mine].reshape(-1).copy()) if mine else torch.zeros(0, dtype=torch.float32)
        send = od.pad_send(local, len(mine), per_rank, per_unit) if mine else torch.zeros(per_rank * per_unit, dtype=torch.float32)
        assert send.numel() == per_rank * per_unit
        if len(mine) == per_rank:
            assert send.data_ptr() == local.data_ptr()          # as it lies: the plan's field buffer is the send buffer
        else:
            assert torch.all(send[len(mine) * per_unit:] == 0)
        chunks.append(send)
    got = torch.cat(chunks)                                     # what all_gather_into_tensor / gather leave on the receiving rank
    if unit_order:
        # the collective may write straight into the result
        np.testing.assert_array_equal(got.numpy().reshape(-1, per_unit)[:n_units], units)
        assert ws == 1 or per_rank == 1
    dst = torch.full((n_units * per_unit,), -1.0)
    od.rank_major_to_unit_order(got, dst, ws, per_rank, per_unit, n_units)
    np.testing.assert_array_equal(dst.numpy().reshape(n_units, per_unit), units)
    # the NumPy model of the same: unit k * ws + r sits at [r, k]
    model = np.full((n_units, per_unit), -1.0, np.float32)
    g = got.numpy().reshape(ws, per_rank, per_unit)
    for r in range(ws):
        for k in range(per_rank):
            if k * ws + r < n_units:
                model[k * ws + r] = g[r, k]
    np.testing.assert_array_equal(model, units)


def test_bench_cpu_placement_plan():
    """bench.py's rank -> CPU set rule (pure arithmetic; the sysfs reads are best effort)."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    nodes = [0, 0, 0, 0, 1, 1, 1, 1]
    node_cpus = {0: set(range(0, 64)) | set(range(128, 192)), 1: set(range(64, 128)) | set(range(192, 256))}
    sets = [bench.plan_affinity(r, 8, nodes, node_cpus, set(range(256))) for r in range(8)]
    assert all(len(s) == 32 for s in sets)
    for r in range(8):
        assert sets[r] <= node_cpus[nodes[r]]
        for q in range(r):
            assert not (sets[r] & sets[q])
    # fewer ranks than GPUs: only the ranks that exist share a node's cores
    assert len(bench.plan_affinity(0, 2, nodes, node_cpus, set(range(256)))) == 64
    assert len(bench.plan_affinity(0, 1, nodes, node_cpus, set(range(256)))) == 128
    # a container that allows 8 cores, unknown topology, too few cores: leave the affinity alone
    assert bench.plan_affinity(0, 1, [], node_cpus, set(range(8))) is None
    assert bench.plan_affinity(0, 1, [-1], node_cpus, set(range(8))) is None
    assert bench.plan_affinity(0, 8, nodes, node_cpus, {0}) is None
    assert isinstance(bench.pin_to_gpu_node(0, 1), str)


def test_bench_self_launch_passes_only_the_json_line(tmp_path, monkeypatch):
    """A child that prints a banner before and after its JSON line (as RCCL / gloo do on the C-level stdout): the parent's stdout is the line."""
    sys.path.insert(0, ROOT)
    fake = tmp_path / "torch" / "distributed"
    fake.mkdir(parents=True)
    (tmp_path / "torch" / "__init__.py").write_text("")
    (fake / "__init__.py").write_text("")
    (fake / "run.py").write_text("import sys\nprint('RCCL version : 2.26.6')\nprint('{\"metric\": \"m\", \"value\": 1.5}')\nprint('[Gloo] Rank 0 is connected')\nsys.exit(3)\n")
    code = ("import sys, os; sys.path.insert(0, %r); sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.self_launch(bench.parse_args(['--gpus', '2']), ['--gpus', '2']))") % (str(tmp_path), ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=dict(os.environ, PYTHONPATH=str(tmp_path)))
    assert r.returncode == 3
    assert r.stdout == '{"metric": "m", "value": 1.5}\n'
    assert "RCCL version" in r.stderr and "Gloo" in r.stderr


def test_bench_self_launch_kills_ranks_that_hang_and_says_where_they_were(tmp_path):
    """a rank launcher that never returns (a hung RCCL initialisation on a node this repository has never seen) costs the
    --launch-timeout budget, not the caller's lease: the child's whole process group is killed -- the ranks it started included --, every rank's last
    log lines are printed, and the exit code is 124."""
    import time
    sys.path.insert(0, ROOT)
    fake = tmp_path / "torch" / "distributed"
    fake.mkdir(parents=True)
    (tmp_path / "torch" / "__init__.py").write_text("")
    (fake / "__init__.py").write_text("")
    pidfile = tmp_path / "rank1.pid"
    (fake / "run.py").write_text(
        "import os, subprocess, sys, time\n"
        "print('[bench rank 0 +0.1s] init_process_group(nccl) ...', file=sys.stderr, flush=True)\n"
        "p = subprocess.Popen([sys.executable, '-c', 'import sys, time; print(\"[bench rank 1 +0.1s] init_process_group(nccl) ...\", file=sys.stderr, flush=True); time.sleep(3600)'])\n"
        f"open({str(pidfile)!r}, 'w').write(str(p.pid))\n"
        "time.sleep(3600)\n")
    code = ("import sys, os; sys.path.insert(0, %r); sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.self_launch(bench.parse_args(['--gpus', '2', '--launch-timeout', '3']), ['--gpus', '2']))") % (str(tmp_path), ROOT)
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=dict(os.environ, PYTHONPATH=str(tmp_path)))
    assert r.returncode == 124, (r.returncode, r.stderr[-2000:])
    assert time.time() - t0 < 40
    assert r.stdout == ""
    assert "did not finish within --launch-timeout 3 s" in r.stderr
    assert "rank 0: [bench rank 0 +0.1s] init_process_group(nccl) ..." in r.stderr and "rank 1: [bench rank 1 +0.1s] init_process_group(nccl) ..." in r.stderr
    pid = int(pidfile.read_text())
    for _ in range(50):                                       # the grandchild (a rank) went with the group
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            break
        # (a zombie still answers kill 0 until it is reaped by init: look at its state)
        try:
            if open(f"/proc/{pid}/stat").read().split(")")[1].split()[0] == "Z":
                break
        except OSError:
            break
        time.sleep(0.1)
    else:
        raise AssertionError("a rank survived the launcher's death")


def test_bench_ranks_keep_stdout_for_the_json_line():
    """guard_stdout(): whatever a library writes to file descriptor 1 afterwards (RCCL's banner is a C-level printf) lands on stderr."""
    code = ("import sys, os; sys.path.insert(0, %r); import bench; bench.guard_stdout(); os.write(1, b'RCCL version : x\\n'); "
            "print('python-level chatter'); bench.emit_json_line({'value': 2})") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout == '{"value": 2}\n'
    assert "RCCL version" in r.stderr and "chatter" in r.stderr


def test_bench_maps_ranks_to_numa_nodes_through_the_visible_devices_mask():
    """A driver that masks or reorders devices (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES): rank r's GPU is the r-th VISIBLE one, and its CPU set follows."""
    import bench
    assert bench.visible_gpu_order(8, {}) == list(range(8))
    assert bench.visible_gpu_order(8, {"HIP_VISIBLE_DEVICES": "4,5,6,7"}) == [4, 5, 6, 7]
    assert bench.visible_gpu_order(8, {"ROCR_VISIBLE_DEVICES": "7,6,5,4", "HIP_VISIBLE_DEVICES": "1,0"}) == [6, 7]
    assert bench.visible_gpu_order(8, {"CUDA_VISIBLE_DEVICES": "2"}) == [2]
    assert bench.visible_gpu_order(8, {"HIP_VISIBLE_DEVICES": "0,9,1"}) == [0]            # the runtime stops at the first invalid entry
    assert bench.visible_gpu_order(8, {"HIP_VISIBLE_DEVICES": "GPU-abc"}) is None
    nodes = [0, 0, 0, 0, 1, 1, 1, 1]
    cpus = {0: set(range(0, 64)), 1: set(range(64, 128))}
    vis = [nodes[k] for k in bench.visible_gpu_order(8, {"HIP_VISIBLE_DEVICES": "4,5"})]
    assert bench.plan_affinity(0, 2, vis, cpus, set(range(128))) == set(range(64, 96))
    assert bench.plan_affinity(1, 2, vis, cpus, set(range(128))) == set(range(96, 128))
