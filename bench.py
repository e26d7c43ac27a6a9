#!/usr/bin/env python3
"""bench.py -- SSFM sample*steps/s on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One bench "step" = one complete propagation of the workload: configuration C2 of SURVEY.md 8
(2^20-sample dual-polarisation complex64 field, FIBER(length=125 km, h=0.125 km) = exactly 1000
SSFM steps; QPSK-like 32 GBd, 16 samples/symbol, 0 dBm/pol; alpha 0.2 dB/km, beta_2 -21.7 ps^2/km,
beta_3 0.13 ps^3/km, gamma 1.3 /(W km)).  The field is resident in HBM when the timed region
starts; each timed step restores the input with a device-to-device copy (16 MiB, ~1e-3 of a
step) so that every step propagates the same physical signal instead of an ever weaker one.

N > 1: configuration C3 of BASELINE.json -- 8 independent 2^20 x 2 fields (WDM channels, seeds 3000..3007), unit i
on rank i % N, each rank's 8/N fields batched in ONE plan; the path shards with no data-path collective
("scaling": "strong": the 8 fields are the whole job).  After the timed region the propagated fields (16 MiB each)
are gathered to rank 0 in GPU memory over RCCL (opticomlib_amd.dist.gather_device) and that is timed separately
(`gather_ms`).  `--workload c3` runs the same 8-field job on one GPU (the N = 1 point of that curve).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOG2N = 20
N_POL = 2
SSFM_STEPS = 1000
LENGTH_KM, H_KM = 125.0, 0.125          # exactly 1000 float32 steps
C3_FIELDS = 8                           # configuration C3: 8 independent WDM channels
HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md chip table (spec)


def cpu_baseline(a, dt, fibre, sample_steps):
    """The oracle (NumPy restatement of the reference, single thread like the reference) timed on
    a bounded sample of the SAME workload: `sample_steps` SSFM steps of the 2^20 x 2 field."""
    from oracle import ssfm_numpy as orc
    t = time.perf_counter()
    orc.fiber_c64(a, dt, length=LENGTH_KM, h=H_KM, max_steps=sample_steps, **fibre)
    el = time.perf_counter() - t
    return {
        "value": a.shape[-1] * sample_steps / el,
        "unit": "sample*steps/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{sample_steps} of {SSFM_STEPS} SSFM steps of the same 2^{LOG2N} x {N_POL} complex64 field, "
                  f"oracle/ssfm_numpy.fiber_c64 (NumPy {np.__version__}), {el:.1f} s, host has {os.cpu_count()} cores",
    }


def secondary_c1(a, dt, fibre, device):
    """Configuration C1 of BASELINE.json (the same 2^20 x 2 field, complex128, 100 x 1 km), reported beside the
    headline C2 figure: measured after the timed region, outside `value`."""
    from opticomlib_amd import _lib, devices
    n = a.shape[-1]
    hs, _ = devices.step_schedule(100.0, 1.0, _lib.C128)
    p = _lib.Plan(n, N_POL, _lib.C128, device=device)
    try:
        p.set_linear_operator(devices.linear_operator(n, dt, fibre["alpha"], fibre["beta_2"], fibre["beta_3"], _lib.C128))
        p.set_field(a)
        p.propagate_fixed(fibre["gamma"], hs)
        p.synchronize()
        reps = 5
        t = time.perf_counter()
        for _ in range(reps):
            p.propagate_fixed(fibre["gamma"], hs)
        p.synchronize()
        el = (time.perf_counter() - t) / reps
    finally:
        p.close()
    rate = n * hs.size / el
    return {"workload": "C1: the same 2^20-sample dual-pol field, complex128, FIBER(length=100, h=1.0) = 100 SSFM steps", "dtype": "c128",
            "value": rate, "unit": "sample*steps/s", "us_per_ssfm_step": el / hs.size * 1e6,
            "step_frac": 2 * N_POL * 16 * rate / (HBM_PEAK_GBS * 1e9),
            "note": "algorithmic bytes 64 B per sample*step (complex128); not part of `value`"}


def _manycore_worker(job):
    seed, dt, fibre, steps = job
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    from opticomlib_amd import workloads
    from oracle import ssfm_numpy as orc
    a = workloads.qpsk_field(1 << LOG2N, seed=seed).astype(np.complex64)
    t = time.perf_counter()
    orc.fiber_c64_tidy(a, dt, length=LENGTH_KM, h=H_KM, max_steps=steps, **fibre)
    return time.perf_counter() - t


def cpu_baseline_manycore(dt, fibre, procs, sample_steps):
    """the tidied NumPy variant, one single-threaded process per independent field
    (how a CPU user would run configurations C3 / C4), `procs` fields at once."""
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(procs) as pool:
        t = time.perf_counter()
        times = pool.map(_manycore_worker, [(3000 + i, dt, fibre, sample_steps) for i in range(procs)])
        wall = time.perf_counter() - t
    return {
        "value": procs * (1 << LOG2N) * sample_steps / max(times), "unit": "sample*steps/s", "cores": procs, "kind": "port",
        "sample": f"{procs} processes x {sample_steps} SSFM steps of independent 2^{LOG2N} x {N_POL} complex64 fields, "
                  f"oracle/ssfm_numpy.fiber_c64_tidy (tabulated exp(D~h), one nonlinear exp per step, in place), slowest worker "
                  f"{max(times):.1f} s, pool wall {wall:.1f} s incl. start-up, host has {os.cpu_count()} cores",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cpu-steps", type=int, default=48, help="SSFM steps of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-profile-pass", action="store_true", help="skip the per-kernel HIP-event pass")
    ap.add_argument("--cpu-manycore", type=int, default=0, help="also time the tidied CPU variant on this many processes (0 = skip)")
    ap.add_argument("--workload", choices=["auto", "c2", "c3"], default="auto", help="auto: C2 (one field) on 1 GPU, C3 (8 fields sharded) on more")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("BENCH_SAME_GPU"):             # diagnostics only: every rank on GPU 0 (needs BENCH_DIST_BACKEND=gloo: RCCL wants one GPU per rank)
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with torch.distributed.run); got {world}")
    distributed = "RANK" in os.environ          # under torch.distributed.run, also with one rank
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")          # diagnostics only: gloo | none
    if backend == "none":
        distributed = False
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from opticomlib_amd import _lib, devices, workloads

    n = 1 << LOG2N
    fibre = dict(workloads.SMF)
    dt = 1.0 / (workloads.BENCH_GV["sps"] * workloads.BENCH_GV["R"])
    c3 = args.workload == "c3" or (args.workload == "auto" and world > 1)
    from opticomlib_amd import dist as od
    if c3:
        units = od.shard(C3_FIELDS, rank, world)                  # round-robin: unit i on rank i % world
        a = np.stack([workloads.qpsk_field(n, seed=3000 + u, n_pol=N_POL) for u in units]) if units else np.zeros((0, N_POL, n))
    else:
        units = [0]
        a = workloads.qpsk_field(n, seed=2024, n_pol=N_POL)[None]                     # C2 seed
    fields_here = len(units)
    hs, _ = devices.step_schedule(LENGTH_KM, H_KM, _lib.C64)
    assert hs.size == SSFM_STEPS

    plan = None
    if fields_here:
        plan = _lib.Plan(n, N_POL * fields_here, _lib.C64, device=local_rank)
        plan.set_linear_operator(devices.linear_operator(n, dt, fibre["alpha"], fibre["beta_2"], fibre["beta_3"], _lib.C64))
        x_dev = torch.from_numpy(np.ascontiguousarray(a.astype(np.complex64))).cuda()       # resident input
    torch.cuda.synchronize()

    def one_step():
        if plan is None:
            return
        plan.set_field_device(x_dev.data_ptr())          # D2D restore on the plan's stream
        plan.propagate_fixed(fibre["gamma"], hs)         # 1 + 2*1000 launches per lane, asynchronous

    def fence():
        if plan is not None:
            plan.synchronize()                           # the plan's own (non-blocking, high-priority) streams
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # plan set-up, not part of the protocol's warm-up: first touch of every buffer and table (and, with
    # SSFM_GRAPH=auto, the library's eager-vs-graph measurement, which needs four runs of the schedule)
    for _ in range(4 if os.environ.get("SSFM_GRAPH", "")[:1] in ("a", "A") else 1):
        one_step()
    fence()
    for _ in range(args.warmup):
        one_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_dev, launches = plan.last_propagate_ms() if plan is not None else (0.0, 0)
    out = plan.get_field().reshape(fields_here, N_POL, n) if plan is not None else np.zeros((0, N_POL, n), np.complex64)
    powers = [float(np.mean(np.abs(o.astype(np.complex128)) ** 2)) for o in out]

    # the only "exchange" of this path: the gather of the results at the end, in GPU memory (RCCL over xGMI)
    gather_ms = None
    checks = powers
    if c3 and distributed and backend == "nccl":
        total = C3_FIELDS
        got = od.gather_device(plan.field_device_ptr if plan is not None else 0, fields_here, (N_POL, n), np.complex64, total, local_rank,
                               to_all=False, owner=plan)                                      # warm-up: communicator set-up
        reps = 3
        fence()
        tg = time.perf_counter()
        for _ in range(reps):
            got = od.gather_device(plan.field_device_ptr if plan is not None else 0, fields_here, (N_POL, n), np.complex64, total, local_rank,
                                   to_all=False, owner=plan)
        fence()
        gather_ms = (time.perf_counter() - tg) / reps * 1e3
        if rank == 0:
            g = got.to_host()
            checks = [float(np.mean(np.abs(g[u].astype(np.complex128)) ** 2)) for u in range(total)]
            mine = od.shard(total, 0, world)
            for k, u in enumerate(mine):                                                      # the gathered block holds this rank's own results in place
                assert np.array_equal(g[u], out[k]), "gathered field differs from the local result"
    elif distributed:
        dev = "cuda" if backend == "nccl" else "cpu"
        gl = [torch.zeros(max(1, -(-C3_FIELDS // world)), dtype=torch.float64, device=dev) for _ in range(world)]
        mine_t = torch.zeros_like(gl[0])
        mine_t[: len(powers)] = torch.tensor(powers, dtype=torch.float64)
        dist.all_gather(gl, mine_t)
        checks = [float(v) for t in gl for v in t.tolist()]

    total_fields = C3_FIELDS if c3 else world
    value = total_fields * n * SSFM_STEPS * args.steps / elapsed                   # whole job: every field of every rank

    roofline = None
    cpu = None
    cpu_many = None
    other = None
    if rank == 0:
        lanes = plan.lanes
        rows_per_launch = N_POL * fields_here // lanes       # a launch covers one lane's rows
        b_alg_launch = 2 * rows_per_launch * 8 * n           # its rows read once + written once
        if not args.no_profile_pass:
            def timed_pass(mode):
                plan.set_profiling(mode)
                one_step()
                plan.synchronize()
                kt = plan.kernel_times()
                plan.set_profiling(0)
                return kt
            sparse = timed_pass(2)                         # cheap: pooled average launch time
            dense = timed_pass(1)                          # per-class split (perturbed by its own events)
            pooled_us = sum(v[1] for v in sparse.values()) / sum(v[0] for v in sparse.values()) * 1e3
            dense_avg = {k: v[1] / max(v[0], 1) for k, v in dense.items()}
            # The committed rocprofv3 --kernel-trace --stats summary of this command: kernel begin -> end only, while the
            # event-based figures additionally contain the dependent-launch gap that follows every kernel on its stream.
            rocprof_us, rocprof_src = {}, None
            for stats_name in ("r02_final_kernel_stats.csv", "r01_final_kernel_stats.csv"):
                stats = os.path.join(ROOT, "profiles", stats_name)
                if not os.path.exists(stats):
                    continue
                try:
                    import csv
                    acc = {}
                    for row in csv.DictReader(open(stats)):
                        for k in dense_avg:
                            if k + "<float" in row["Name"]:          # the C2 kernels (the file also holds C1's complex128 ones)
                                c, tns = acc.get(k, (0, 0.0))
                                acc[k] = (c + int(row["Calls"]), tns + float(row["TotalDurationNs"]))
                    rocprof_us = {k: tns / c / 1e3 for k, (c, tns) in acc.items() if c}
                    rocprof_src = f"profiles/{stats_name} (kernel begin->end, no launch gap)"
                except Exception:
                    pass
                break
            # Split of the pooled time between the two kernels.  An event after every launch doubles the host's work per
            # launch; once the kernels are faster than that, the event-to-event intervals only show the host's cadence
            # (both classes come out equal to four digits) and carry no information: then the split follows the ratio of
            # the committed rocprofv3 averages, so that `kernel` is the same kernel in both sources.
            ks = sorted(dense_avg)
            degenerate = len(ks) == 2 and abs(dense_avg[ks[0]] / max(dense_avg[ks[1]], 1e-12) - 1.0) < 0.01
            if degenerate and len(rocprof_us) == 2:
                mean_r = sum(rocprof_us.values()) / 2
                launch_us = {k: pooled_us * rocprof_us[k] / mean_r for k in ks}
                split_source = "ratio of the committed rocprofv3 averages (the event-per-launch pass was host-bound: equal intervals)"
            else:
                dense_pooled = sum(v[1] for v in dense.values()) / sum(v[0] for v in dense.values())
                launch_us = {k: pooled_us * dense_avg[k] / dense_pooled for k in dense}
                split_source = "event after every launch"
            dom = max(launch_us, key=launch_us.get)
            avg_us = launch_us[dom]
            achieved = b_alg_launch / (avg_us * 1e-6) / 1e9
            roofline = {
                "bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "avg_launch_us": avg_us,
                "launch_us": launch_us,
                "launch_us_split_source": split_source,
                "pooled_launch_us": pooled_us,
                "launch_us_dense_events": {k: v * 1e3 for k, v in dense_avg.items()},
                "rocprof_kernel_us": rocprof_us or None,
                "rocprof_avg_us_of_kernel": rocprof_us.get(dom),
                "rocprof_source": rocprof_src,
                "algorithmic_bytes_per_launch": b_alg_launch,
                "lanes": lanes, "rows_per_launch": rows_per_launch,
                "note": "HIP events on the launch's own stream. pooled_launch_us: one event per 64 launches "
                        "(interval / launches, includes the dependent-launch gap of ~1.5 us that rocprofv3's begin->end "
                        "durations do not). launch_us: the pooled time split between the two kernels (launch_us_split_source). "
                        "With lanes > 1 launches of different row groups overlap on the chip; the chip-level figure is step_frac",
                "step_frac": 2 * N_POL * 8 * (value * fields_here / total_fields) / (HBM_PEAK_GBS * 1e9),
            }
            for pmc_name in ("r02_pmc_traffic.json", "pmc_traffic.json"):
                pmc = os.path.join(ROOT, "profiles", pmc_name)
                if not os.path.exists(pmc):
                    continue
                try:
                    t = json.load(open(pmc))
                    # (the PMC passes were taken on one-row launches; a launch over more rows moves that many times the field bytes)
                    per_row = {k: t.get(k, {}).get("bytes_per_launch") for k in launch_us}
                    roofline["traffic"] = per_row.get(dom) * rows_per_launch if per_row.get(dom) else None
                    roofline["traffic_per_kernel"] = {k: (v * rows_per_launch if v else None) for k, v in per_row.items()}
                    roofline["traffic_source"] = f"profiles/{pmc_name} (per one-row launch, x rows_per_launch): " + str(t.get("_source"))
                except Exception:
                    pass
                break
        if world == 1 and not c3:
            other = secondary_c1(a[0], dt, fibre, local_rank)
        if world == 1 and args.cpu_steps > 0:
            cpu = cpu_baseline(a[0], dt, fibre, args.cpu_steps)
        if world == 1 and args.cpu_manycore > 0:
            cpu_many = cpu_baseline_manycore(dt, fibre, args.cpu_manycore, max(8, args.cpu_steps // 4))

    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    print(json.dumps({
        "metric": "SSFM sample*steps/sec, 2^20-sample dual-pol fiber",
        "value": value,
        "unit": "sample*steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if c3 else "weak",
        "vs_baseline": None,
        "dtype": "c64",
        "data": "synthetic",
        "config": {
            "workload": (f"C3: {C3_FIELDS} independent 2^20-sample dual-pol fields (WDM channels, seeds 3000..), FIBER(length=125, h=0.125) = 1000 SSFM steps each, "
                         f"complex64, unit i on rank i % {world}, a rank's fields batched in one plan" if c3 else
                         "C2: 2^20-sample dual-pol optical_signal, FIBER(length=125, h=0.125) = 1000 SSFM steps, complex64"),
            "n_samples": n, "n_pol": N_POL, "ssfm_steps_per_bench_step": SSFM_STEPS,
            "fields_total": total_fields, "fields_per_gpu": fields_here, "parallelism": f"independent-fields x{world}",
        },
        **({"gather_ms": gather_ms, "gather_bytes": C3_FIELDS * N_POL * n * 8,
            "gather_note": "all propagated fields to rank 0 in GPU memory, one RCCL gather on the plans' field buffers; after the timed region, not in `value`"}
           if gather_ms is not None else {}),
        "device_ms_last_propagate": ms_dev,
        "launches_per_propagate": launches,
        "us_per_ssfm_step": elapsed / args.steps / SSFM_STEPS * 1e6,
        "output_power_W_per_rank": checks,
        "roofline": roofline,
        "cpu_baseline": cpu,
        **({"cpu_baseline_manycore": cpu_many} if cpu_many else {}),
        **({"secondary": other} if other else {}),
    }))


if __name__ == "__main__":
    main()
