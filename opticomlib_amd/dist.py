"""Sharding of INDEPENDENT propagations over the GPUs of one node.

The reference has no parallelism of any kind; what shards naturally is the outer loop a user
writes around ``FIBER``: one call per WDM channel (configuration C3) or per Monte-Carlo PRBS
realisation (C4).  Those calls never exchange data, so there is NO collective on the data path:
unit ``i`` goes to rank ``i % world``, every rank runs its units on its own GPU, and ONE gather at
the end (RCCL over xGMI when the backend is ``nccl``; ``gloo`` on CPU in the tests) returns the
results in unit order.  xGMI is point-to-point, so the gather is a single all-gather of equal
chunks -- 16 MiB per 2^20 x 2 complex64 field -- not a ring of small messages.

One process per GPU: launch with ``python -m torch.distributed.run --nproc-per-node N ...``; the
device is ``LOCAL_RANK`` (``devices.default_device``).
"""
from __future__ import annotations

import os
from typing import Callable, List, Sequence

import numpy as np


def world():
    """(rank, world_size) of this process; (0, 1) when torch.distributed is not initialised."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    return 0, 1


def init(backend: str | None = None):
    """Initialise the default process group from the torchrun environment (idempotent)."""
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return world()
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return 0, 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        kw["device_id"] = torch.device("cuda", local)
    dist.init_process_group(backend, **kw)
    return world()


def shard(n_units: int, rank: int | None = None, world_size: int | None = None) -> List[int]:
    """Unit indices owned by ``rank``: round-robin, so any prefix of the units is balanced."""
    if rank is None or world_size is None:
        rank, world_size = world()
    return list(range(rank, n_units, world_size))


def gather_results(local: Sequence[np.ndarray], n_units: int, to_all: bool = True):
    """Gather per-unit arrays (all of one shape and dtype) into unit order.

    ``local[k]`` is the result of unit ``shard(n_units)[k]``.  Returns a list of ``n_units``
    arrays on every rank (``to_all``) or on rank 0 only (others get ``None``).
    """
    rank, ws = world()
    if ws == 1:
        return list(local)
    import torch
    import torch.distributed as dist
    per_rank = -(-n_units // ws)                       # ceil: ranks with fewer units pad
    first = np.asarray(local[0]) if len(local) else None
    # every rank owns at least one unit when n_units >= world; otherwise learn the shape from rank 0
    meta = [None]
    if rank == 0:
        meta = [(first.shape, first.dtype.str)]
    dist.broadcast_object_list(meta, src=0)
    shape, dtstr = meta[0]
    dt = np.dtype(dtstr)
    buf = np.zeros((per_rank,) + tuple(shape), dtype=dt)
    for k, a in enumerate(local):
        buf[k] = a
    # complex arrays travel as real pairs (NCCL has no complex types)
    view = buf.view(np.float64 if dt == np.complex128 else np.float32) if dt.kind == "c" else buf
    use_cuda = dist.get_backend() == "nccl"
    t = torch.from_numpy(np.ascontiguousarray(view))
    if use_cuda:
        t = t.cuda()
    if to_all:
        outs = [torch.empty_like(t) for _ in range(ws)]
        dist.all_gather(outs, t)
    else:
        outs = [torch.empty_like(t) for _ in range(ws)] if rank == 0 else None
        dist.gather(t, outs, dst=0)
        if rank != 0:
            return None
    res: List[np.ndarray] = [None] * n_units
    for r in range(ws):
        arr = outs[r].cpu().numpy()
        if dt.kind == "c":
            arr = arr.view(dt)
        arr = arr.reshape((per_rank,) + tuple(shape))
        for k, unit in enumerate(range(r, n_units, ws)):
            res[unit] = arr[k]
    return res


def sharded_map(fn: Callable[[int], np.ndarray], n_units: int, to_all: bool = True):
    """Run ``fn(unit)`` for this rank's units, then gather: the whole multi-GPU path."""
    mine = shard(n_units)
    local = [np.asarray(fn(u)) for u in mine]
    return gather_results(local, n_units, to_all=to_all)


def propagate_channels(fields, dt: float, to_all: bool = True, dbp: bool = False, **fiber_kw):
    """``FIBER`` on every field of ``fields`` (shape ``(F, [2,] N)``), sharded over the ranks; with
    ``dbp=True`` followed by ``DBP`` with the same parameters (configuration C4: forward + digital
    back-propagation per Monte-Carlo realisation) without the field leaving the GPU in between.

    Each field is an independent propagation, exactly as F separate ``FIBER`` calls of the reference
    would be.  With a fixed step ``h`` the rows of different fields never interact, so all fields of
    this rank are propagated together in ONE plan (their rows are just more batch rows; several
    fields resident on a GPU run at 15.5 instead of 21 us per field-step).  With the adaptive step
    (``h=None``) every field keeps its own step-size sequence, so they run one after the other.
    Returns the list of F output arrays (complex64, or complex128 with ``precision="complex128"``).
    """
    from . import _lib
    from .devices import FIBER, _check_size, _is_fast_size, _precision_code, get_plan, linear_operator, step_schedule
    from .typing import gv, optical_signal

    fields = np.asarray(fields)
    n_units = fields.shape[0]
    mine = shard(n_units)
    kw = dict(fiber_kw)
    fixed = kw.get("h") is not None and not kw.get("return_steps") and not kw.get("show_progress")
    fixed = fixed and _is_fast_size(fields.shape[-1], _precision_code(kw.get("precision", "complex64")))     # other lengths: one by one (chirp-z)
    local: List[np.ndarray] = []
    if fixed and len(mine) > 1:
        prec = _precision_code(kw.get("precision", "complex64"))
        cdt = np.complex64 if prec == _lib.C64 else np.complex128
        stack = np.ascontiguousarray(fields if len(mine) == n_units else fields[mine], dtype=cdt)     # one process: no copy
        unit_shape = stack.shape[1:]
        n = unit_shape[-1]
        _check_size(n, prec)
        rows = stack.reshape(-1, n)
        plan = get_plan(n, rows.shape[0], prec, kw.get("device"))
        plan.set_linear_operator(linear_operator(n, dt, kw.get("alpha", 0.0), kw.get("beta_2", 0.0), kw.get("beta_3", 0.0), prec))
        plan._op_key = None
        plan.set_field(rows)
        hs, _ = step_schedule(kw["length"], kw["h"], prec)
        if hs.size:
            plan.propagate_fixed(kw.get("gamma", 0.0), hs)
            if dbp:                                   # DBP = FIBER with negated parameters (devices.py:1280-1283)
                plan.set_linear_operator(linear_operator(n, dt, -kw.get("alpha", 0.0), -kw.get("beta_2", 0.0), -kw.get("beta_3", 0.0), prec))
                plan.propagate_fixed(-kw.get("gamma", 0.0), hs)
        out = plan.get_field().reshape((len(mine),) + unit_shape)
        local = [out[k] for k in range(len(mine))]
    else:
        from .devices import DBP
        for u in mine:
            gv.dt, gv.fs = dt, 1.0 / dt
            y = FIBER(optical_signal(fields[u]), **kw)
            local.append((DBP(y, **kw) if dbp else y).signal)
    return gather_results(local, n_units, to_all=to_all)
