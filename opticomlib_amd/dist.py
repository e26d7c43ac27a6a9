"""Sharding of INDEPENDENT propagations over the GPUs of one node.

The reference has no parallelism of any kind; what shards naturally is the outer loop a user
writes around ``FIBER``: one call per WDM channel (configuration C3) or per Monte-Carlo PRBS
realisation (C4).  Those calls never exchange data, so there is NO collective on the data path:
unit ``i`` goes to rank ``i % world``, every rank runs its units on its own GPU, and ONE gather at
the end returns the results in unit order.

The gather works on DEVICE memory: each rank's results already lie side by side in GPU memory (the
plan's field buffer of a batched run, or the device arrays the calls returned); they are handed to
RCCL where they lie (``__cuda_array_interface__``, no copy), one ``all_gather_into_tensor`` (or
``gather`` to rank 0) moves them over xGMI -- point-to-point links, so one collective of equal
chunks, 16 MiB per 2^20 x 2 complex64 field, not a ring of small messages -- and the consumer
downloads the gathered block ONCE (or keeps it on the GPU: ``on_device=True``).  With the ``gloo``
backend (CPU tests of the harness) the same functions gather host arrays.

One process per GPU: launch with ``python -m torch.distributed.run --nproc-per-node N ...``; the
device is ``LOCAL_RANK`` (``devices.default_device``).
"""
from __future__ import annotations

import contextlib
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
    if "RANK" not in os.environ:
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


def _device_backend() -> bool:
    import torch.distributed as dist
    return dist.is_initialized() and dist.get_backend() == "nccl"


class _RawDeviceView:
    """A raw device pointer as a real-valued CUDA array (what torch / RCCL can take without a copy)."""

    def __init__(self, ptr: int, count: int, real_dtype, owner=None):
        self._owner = owner                                   # keeps the memory alive
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": np.dtype(real_dtype).str, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def _torch_view(ptr: int, count: int, real_dtype, device: int, owner=None):
    """1-D real torch tensor over ``count`` numbers at device address ``ptr`` (no copy)."""
    import torch
    return torch.as_tensor(_RawDeviceView(ptr, count, real_dtype, owner), device=torch.device("cuda", device))


def _real_dtype(dt: np.dtype):
    dt = np.dtype(dt)
    return (np.dtype(np.float64) if dt == np.complex128 else np.dtype(np.float32)) if dt.kind == "c" else dt


# how many collectives this process has really issued on device memory (tests assert that the one-rank RCCL run is not a shortcut)
COLLECTIVES = {"all_gather_into_tensor": 0, "gather": 0}


# ---- the arithmetic of gather_device, on plain torch tensors (any device): what tests/test_dist_cpu.py runs for world sizes no box has offered yet
def gather_layout(n_units: int, ws: int):
    """(units per rank incl. padding, whether the collective's rank-major receive buffer IS the result in unit order)."""
    per_rank = -(-n_units // ws)                                                    # ceil: ranks with fewer units pad
    in_place = ws * per_rank == n_units and (ws == 1 or per_rank == 1)
    return per_rank, in_place


def pad_send(local, n_local: int, per_rank: int, per_unit: int):
    """A rank's send buffer: its ``n_local`` units as they lie (no copy when the rank owns ``per_rank`` of them), zero padded otherwise."""
    import torch
    if n_local == per_rank:
        return local
    send = torch.zeros(per_rank * per_unit, dtype=local.dtype, device=local.device)
    if n_local:
        send[: n_local * per_unit].copy_(local[: n_local * per_unit])
    return send


def rank_major_to_unit_order(got, dst, ws: int, per_rank: int, per_unit: int, n_units: int):
    """``got[r, k]`` is unit ``k * ws + r`` (unit i lives on rank i % ws): one strided copy into unit order, the padding dropped."""
    dst.view(n_units, per_unit).copy_(got.view(ws, per_rank, per_unit).transpose(0, 1).reshape(ws * per_rank, per_unit)[:n_units])
    return dst


def gather_device(local_ptr: int, n_local: int, unit_shape, dtype, n_units: int, device: int, to_all: bool = True, owner=None,
                  force_collective: bool = True):
    """Gather per-unit results that lie side by side in the memory of GPU ``device`` (``n_local`` units of
    ``unit_shape`` and ``dtype`` at ``local_ptr``; unit k is unit ``shard(n_units)[k]``) into ONE device array of
    shape ``(n_units,) + unit_shape`` in unit order: ``_lib.DeviceArray`` on every rank (``to_all``) or on rank 0
    (others get ``None``).  One collective on device memory; nothing crosses PCIe.

    The collective runs whenever a device process group exists, ALSO with one rank (``force_collective``; a
    one-rank RCCL run is how the path is exercised on a one-GPU box).  The receive buffer is the result itself
    whenever rank-major order is unit order (one rank, or one unit per rank -- configuration C3 on 8 GPUs): RCCL
    writes the chunks where they belong; otherwise one strided device copy reorders them.  Every rank returns only
    after the collective has completed on ITS device: the send buffer is usually a plan's field buffer, which the
    plan's own stream is free to overwrite as soon as the caller lets go of the plan."""
    import torch
    import torch.distributed as dist
    from . import _lib
    rank, ws = world()
    dtype = np.dtype(dtype)
    rdt = _real_dtype(dtype)
    per_unit = int(np.prod(unit_shape)) * (2 if dtype.kind == "c" else 1)          # real numbers per unit
    per_rank, unit_order = gather_layout(n_units, ws)
    dev = torch.device("cuda", device)
    tdt = torch.float64 if rdt == np.float64 else torch.float32
    if n_local:
        send = pad_send(_torch_view(local_ptr, n_local * per_unit, rdt, device, owner), n_local, per_rank, per_unit)      # as it lies when n_local == per_rank
    else:
        send = torch.zeros(per_rank * per_unit, dtype=tdt, device=dev)
    need = to_all or rank == 0
    collective = _device_backend() and (ws > 1 or force_collective)
    in_place = need and unit_order                                                  # rank-major order IS unit order
    out = _lib.DeviceArray((n_units,) + tuple(unit_shape), dtype, device) if need else None
    dst = _torch_view(out.ptr, n_units * per_unit, rdt, device, out) if need else None
    got = None
    if need:
        got = dst if in_place else torch.empty(ws * per_rank * per_unit, dtype=tdt, device=dev)
    if not collective:
        if ws != 1:
            raise RuntimeError("gather_device needs a device (nccl) process group for more than one rank")
        dst.copy_(send)                      # no process group (or the collective declined): one rank's units are the result
    elif to_all:
        dist.all_gather_into_tensor(got, send)
        COLLECTIVES["all_gather_into_tensor"] += 1
    else:
        # rank 0 receives every chunk where it belongs: views of the one receive buffer, no list of temporaries
        dist.gather(send, list(got.chunk(ws)) if rank == 0 else None, dst=0)
        COLLECTIVES["gather"] += 1
    if need and not in_place:
        rank_major_to_unit_order(got, dst, ws, per_rank, per_unit, n_units)       # (one strided device copy)
    torch.cuda.synchronize(dev)              # on EVERY rank: the collective has read `send` (a plan's field buffer) when we return
    return out


def gather_results(local: Sequence, n_units: int, to_all: bool = True, on_device: bool = False):
    """Gather per-unit arrays (all of one shape and dtype) into unit order.

    ``local[k]`` is the result of unit ``shard(n_units)[k]``: NumPy arrays, or ``_lib.DeviceArray`` s.  Device arrays
    under the ``nccl`` backend are gathered in GPU memory (:func:`gather_device`); the result is downloaded once
    (list of ``n_units`` NumPy arrays) unless ``on_device``.  Returns the list on every rank (``to_all``) or on rank 0
    only (others get ``None``).
    """
    from . import _lib
    rank, ws = world()
    is_dev = len(local) > 0 and all(isinstance(a, _lib.DeviceArray) for a in local)
    if ws == 1 and not (is_dev and _device_backend()):       # (a one-rank device group still runs the collective below)
        if on_device or not is_dev:
            return list(local)
        return [a.to_host() for a in local]
    import torch
    import torch.distributed as dist
    if _device_backend():
        from .devices import default_device
        dev = local[0].device if is_dev else default_device()
        meta = [None]
        if rank == 0:
            first = local[0]
            meta = [(tuple(first.shape), np.dtype(first.dtype).str)]
        dist.broadcast_object_list(meta, src=0, device=torch.device("cuda", dev))
        shape, dtstr = meta[0]
        dt = np.dtype(dtstr)
        # this rank's units side by side in one device block
        block = _lib.DeviceArray((max(len(local), 1),) + tuple(shape), dt, dev)
        nb = int(np.prod(shape)) * dt.itemsize
        for k, a in enumerate(local):
            d = a if isinstance(a, _lib.DeviceArray) else _lib.DeviceArray.from_host(np.asarray(a), dt, dev)
            _lib._check(_lib.load().ssfm_device_copy(dev, _lib._VP(block.ptr + k * nb), _lib._VP(d.ptr), nb, 2), "ssfm_device_copy")
        out = gather_device(block.ptr, len(local), shape, dt, n_units, dev, to_all=to_all, owner=block)
        if out is None:
            return None
        if on_device:
            return out
        host = out.to_host()
        return [host[u] for u in range(n_units)]
    # host arrays over gloo (the CPU tests of the harness)
    local = [a.to_host() if isinstance(a, _lib.DeviceArray) else np.asarray(a) for a in local]
    per_rank = -(-n_units // ws)
    meta = [None]
    if rank == 0:
        meta = [(local[0].shape, local[0].dtype.str)]
    dist.broadcast_object_list(meta, src=0)
    shape, dtstr = meta[0]
    dt = np.dtype(dtstr)
    buf = np.zeros((per_rank,) + tuple(shape), dtype=dt)
    for k, a in enumerate(local):
        buf[k] = a
    view = buf.view(_real_dtype(dt)) if dt.kind == "c" else buf        # complex arrays travel as real pairs
    t = torch.from_numpy(np.ascontiguousarray(view))
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
        arr = outs[r].numpy()
        if dt.kind == "c":
            arr = arr.view(dt)
        arr = arr.reshape((per_rank,) + tuple(shape))
        for k, unit in enumerate(range(r, n_units, ws)):
            res[unit] = arr[k]
    return res


def sharded_map(fn: Callable[[int], np.ndarray], n_units: int, to_all: bool = True):
    """Run ``fn(unit)`` for this rank's units, then gather: the whole multi-GPU path."""
    from . import _lib
    mine = shard(n_units)
    local = [fn(u) for u in mine]
    local = [a if isinstance(a, _lib.DeviceArray) else np.asarray(a) for a in local]
    return gather_results(local, n_units, to_all=to_all)


@contextlib.contextmanager
def _sampling_grid(dt: float):
    """The global sampling grid set to ``dt`` for the calls inside, and put back afterwards (``FIBER`` reads
    ``gv.dt`` at call time, reference ``typing.py:1641``; the caller's grid is not ours to change)."""
    from .typing import gv
    saved = (gv.dt, gv.fs)
    gv.dt, gv.fs = dt, 1.0 / dt
    try:
        yield
    finally:
        gv.dt, gv.fs = saved


def propagate_channels(fields, dt: float, to_all: bool = True, dbp: bool = False, on_device: bool = False, **fiber_kw):
    """``FIBER`` on every field of ``fields`` (shape ``(F, [2,] N)``), sharded over the ranks; with
    ``dbp=True`` followed by ``DBP`` with the same parameters (configuration C4: forward + digital
    back-propagation per Monte-Carlo realisation) without the field leaving the GPU in between.

    Each field is an independent propagation, exactly as F separate ``FIBER`` calls of the reference
    would be.  With a fixed step ``h`` the rows of different fields never interact, so all fields of
    this rank are propagated together in ONE plan (their rows are just more batch rows; several
    fields resident on a GPU run at 15 instead of 21 us per field-step).  With the adaptive step
    (``h=None``) every field keeps its own step-size sequence, so they run one after the other.
    The results are gathered in GPU memory (module docstring) and downloaded once: a list of F output
    arrays (complex64, or complex128 with ``precision="complex128"``) on every rank (``to_all``) or on
    rank 0; with ``on_device`` one ``_lib.DeviceArray`` of shape ``(F, [2,] N)`` instead.
    ``fields`` may also be a ``_lib.DeviceArray`` holding THIS rank's units (generated on the device).
    """
    from . import _lib
    from .devices import DBP, FIBER, _check_size, _is_fast_size, _precision_code, default_device, get_plan, linear_operator, step_schedule
    from .typing import optical_signal

    kw = dict(fiber_kw)
    prec = _precision_code(kw.get("precision", "complex64"))
    cdt = np.dtype(np.complex64 if prec == _lib.C64 else np.complex128)
    dev = default_device() if kw.get("device") is None else int(kw["device"])
    rank, ws = world()
    local_dev = isinstance(fields, _lib.DeviceArray)             # this rank's units, already in GPU memory
    if local_dev:
        n_mine = fields.shape[0]
        counts = [n_mine]
        if ws > 1:
            import torch
            import torch.distributed as dist
            t = torch.tensor([n_mine], device=torch.device("cuda", dev) if _device_backend() else "cpu")
            g = [torch.zeros_like(t) for _ in range(ws)]
            dist.all_gather(g, t)
            counts = [int(x.item()) for x in g]
        n_units = sum(counts)
        if counts != [len(shard(n_units, r, ws)) for r in range(ws)]:
            raise ValueError(f"device-resident units must follow the round-robin partition: got {counts} units per rank")
        mine = shard(n_units)
        unit_shape = tuple(fields.shape[1:])
    else:
        fields = np.asarray(fields)
        n_units = fields.shape[0]
        mine = shard(n_units)
        unit_shape = tuple(fields.shape[1:])
    n = unit_shape[-1]
    fixed = kw.get("h") is not None and not kw.get("return_steps") and not kw.get("show_progress")
    fixed = fixed and _is_fast_size(n, prec)                      # other lengths: one by one (chirp-z)
    if fixed and len(mine) == 0:
        # a rank without units (more ranks than units) still takes part in the collective, with nothing to send
        if ws > 1 and _device_backend():
            out = gather_device(0, 0, unit_shape, cdt, n_units, dev, to_all=to_all)
            if out is None:
                return None
            if on_device:
                return out
            host = out.to_host()
            return [host[u] for u in range(n_units)]
        return gather_results([], n_units, to_all=to_all)
    if fixed:
        _check_size(n, prec)
        rows = len(mine) * (int(np.prod(unit_shape[:-1])) if len(unit_shape) > 1 else 1)
        plan = get_plan(n, rows, prec, dev)
        with plan.lock:
            plan.set_linear_operator(linear_operator(n, dt, kw.get("alpha", 0.0), kw.get("beta_2", 0.0), kw.get("beta_3", 0.0), prec))
            if local_dev:
                src = fields if fields.dtype == cdt else fields.astype(cdt)
                plan.set_field_device(src.ptr)
            else:
                stack = np.ascontiguousarray(fields if len(mine) == n_units else fields[mine], dtype=cdt)     # one process: no copy
                plan.set_field(stack.reshape(rows, n))
            hs, _ = step_schedule(kw["length"], kw["h"], prec)
            if hs.size:
                plan.propagate_fixed(kw.get("gamma", 0.0), hs)
                if dbp:                                   # DBP = FIBER with negated parameters (devices.py:1280-1283)
                    plan.set_linear_operator(linear_operator(n, dt, -kw.get("alpha", 0.0), -kw.get("beta_2", 0.0), -kw.get("beta_3", 0.0), prec))
                    plan.propagate_fixed(-kw.get("gamma", 0.0), hs)
            plan.synchronize()
            if _device_backend():
                # the plan's field buffer IS the send buffer (one rank included: the collective still runs)
                out = gather_device(plan.field_device_ptr, len(mine), unit_shape, cdt, n_units, dev, to_all=to_all, owner=plan)
            else:
                out = _lib.DeviceArray((len(mine),) + unit_shape, cdt, dev)
                plan.get_field_device(out.ptr)
                plan.synchronize()
        if ws > 1 and not _device_backend():
            return gather_results([out.to_host()[k] for k in range(len(mine))], n_units, to_all=to_all)
        if out is None:
            return None
        if on_device:
            return out
        host = out.to_host()
        return [host[u] for u in range(n_units)]
    local = []
    kw.pop("device", None)
    with _sampling_grid(dt):
        for k, u in enumerate(mine):
            if local_dev:
                nb = int(np.prod(unit_shape)) * fields.dtype.itemsize
                d = _lib.DeviceArray(unit_shape, fields.dtype, dev)
                _lib._check(_lib.load().ssfm_device_copy(dev, _lib._VP(d.ptr), _lib._VP(fields.ptr + k * nb), nb, 2), "ssfm_device_copy")
                x = optical_signal.from_device(d)
            else:
                x = optical_signal(fields[u])
            y = FIBER(x, device=dev, **kw)
            if dbp:
                y = DBP(y, device=dev, **kw)
            local.append(y._raw("signal"))
    if ws == 1 and on_device and not _device_backend():
        blk = _lib.DeviceArray((n_units,) + unit_shape, cdt, dev)
        nb = int(np.prod(unit_shape)) * cdt.itemsize
        for k, a in enumerate(local):
            d = a if isinstance(a, _lib.DeviceArray) else _lib.DeviceArray.from_host(np.asarray(a), cdt, dev)
            _lib._check(_lib.load().ssfm_device_copy(dev, _lib._VP(blk.ptr + k * nb), _lib._VP(d.ptr), nb, 2), "ssfm_device_copy")
        return blk
    return gather_results(local, n_units, to_all=to_all, on_device=on_device)
