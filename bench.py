#!/usr/bin/env python3
"""bench.py -- SSFM sample*steps/s on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload auto|c2|c3|c4]

One bench "step" = one complete pass of the workload over its synthetic input, resident in HBM when the timed region
starts (each timed step restores the input with a device-to-device copy, ~1e-3 of a step, so that every step propagates
the same physical signal instead of an ever weaker one).

  auto / c2   configuration C2 of SURVEY.md 8 on EVERY GPU: a 2^20-sample dual-polarisation complex64 field,
              FIBER(length=125 km, h=0.125 km) = exactly 1000 SSFM steps; QPSK-like 32 GBd, 16 samples/symbol, 0 dBm/pol;
              alpha 0.2 dB/km, beta_2 -21.7 ps^2/km, beta_3 0.13 ps^3/km, gamma 1.3 /(W km).  N = 1: seed 2024 (the headline
              line).  N > 1: one such field per rank, seeds 3000 + rank -- independent WDM channels, "scaling": "weak"; at
              N = 8 this IS configuration C3 (8 channels, one per GPU).
  c3          configuration C3 as a fixed job: 8 channels (seeds 3000..3007), unit i on rank i % N, a rank's channels batched in
              ONE plan ("scaling": "strong").
  c4          configuration C4: 64 Monte-Carlo PRBS realisations (LFSR seeds 1..64, generated ON the device), FIBER(100 x 1 km)
              then DBP(100 x 1 km) back to back in GPU memory, unit i on rank i % N, 8 realisations resident at a time.

The path shards over independent fields: NO collective inside the timed region.  With a process group the propagated
fields are then gathered to rank 0 in GPU memory by ONE RCCL collective on the plans' field buffers
(opticomlib_amd.dist.gather_device), timed separately as `gather_ms`.

Launching.  `--gpus N` with N > 1 (or `--workload c3|c4` at any N, so that the RCCL gather really runs) and no RANK in the
environment: this process -- BEFORE it imports torch or touches a GPU -- starts
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same args>
as a child, passes on the ONE JSON line of the child's rank 0 and exits with the child's code.  Under torch.distributed.run itself
(RANK set) it is one of the ranks.  Prints ONE JSON line on rank 0.

stdout carries the JSON line and nothing else: every rank points file descriptor 1 at stderr before any library is loaded (RCCL prints a
five-line version banner and gloo its connection report on the C-level stdout) and keeps a private duplicate of the real stdout for the
line; a self-launching parent also filters its child's stdout down to the last JSON line.

CPU placement: before anything touches a GPU every rank pins itself (os.sched_setaffinity) to the cores of its GPU's NUMA node
(/sys/class/drm/card*/device/numa_node -> /sys/devices/system/node/nodeK/cpulist), split between the ranks that share the node; the set is
reported as `cpu_affinity`.  A two-lane run is sensitive to the host's enqueue rate.  BENCH_NO_PIN=1 leaves the affinity alone.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOG2N = 20
N_POL = 2
SSFM_STEPS = 1000
LENGTH_KM, H_KM = 125.0, 0.125          # exactly 1000 float32 steps
C3_FIELDS = 8                           # configuration C3: 8 independent WDM channels
C4_SEEDS = 64                           # configuration C4: 64 PRBS realisations, FIBER + DBP
C4_LENGTH_KM, C4_H_KM = 100.0, 1.0      # 100 steps each way
C4_RESIDENT = 8                         # realisations propagated together in one plan
HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md chip table (spec)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cpu-steps", type=int, default=48, help="SSFM steps of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-profile-pass", action="store_true", help="skip the per-kernel HIP-event passes")
    ap.add_argument("--no-secondary", action="store_true", help="skip the C1 (complex128) figure reported beside the headline")
    ap.add_argument("--no-big-field", action="store_true", help="skip the 2^24 x 2 figure (its split plan runs the headline's column kernel over 32 rows per launch: under "
                                                                "rocprofv3 --stats those launches would share a row of the summary with the headline's)")
    ap.add_argument("--cpu-manycore", type=int, default=0, help="also time the tidied CPU variant on this many processes (0 = skip)")
    ap.add_argument("--workload", choices=["auto", "c2", "c3", "c4"], default="auto")
    ap.add_argument("--launch-timeout", type=float, default=600.0,
                    help="wall-clock budget [s] of a self-launched multi-rank run: beyond it the ranks' process group is killed, what every rank last "
                         "logged is printed, and the exit code is 124 (0 = no limit)")
    return ap.parse_args(argv)


_REAL_STDOUT = None


def guard_stdout():
    """fd 1 -> stderr for everything that is not the JSON line (C libraries included); returns nothing, remembers the real stdout."""
    global _REAL_STDOUT
    if _REAL_STDOUT is not None:
        return
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    sys.stdout = os.fdopen(os.dup(2), "w", buffering=1)          # Python-level prints of imported modules follow


def emit_json_line(obj):
    line = (json.dumps(obj) + "\n").encode()
    fd = _REAL_STDOUT if _REAL_STDOUT is not None else 1
    while line:
        line = line[os.write(fd, line):]


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_nodes():
    """NUMA node of every GPU in HIP's enumeration order, -1 where the kernel does not say.  HIP numbers the GPUs in the order of the KFD
    topology nodes (/sys/class/kfd/kfd/topology/nodes/N/properties: simd_count > 0, PCI domain + location_id); without that directory:
    PCI address order of the amdgpu DRM cards."""
    import glob
    nodes = []
    kfd = []
    for pth in glob.glob("/sys/class/kfd/kfd/topology/nodes/[0-9]*/properties"):
        try:
            props = dict(line.split()[:2] for line in open(pth).read().splitlines() if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) <= 0:
                continue
            loc, dom = int(props["location_id"]), int(props.get("domain", "0"))
            addr = f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7:x}"
            kfd.append((int(os.path.basename(os.path.dirname(pth))), addr))
        except (OSError, ValueError, KeyError):
            continue
    if kfd:
        for _, addr in sorted(kfd):
            try:
                nodes.append(int(open(f"/sys/bus/pci/devices/{addr}/numa_node").read().strip()))
            except (OSError, ValueError):
                nodes.append(-1)
        return nodes
    cards = []
    for dev in glob.glob("/sys/class/drm/card[0-9]*/device"):
        if os.path.basename(os.path.dirname(dev)).count("-"):
            continue                                             # connectors (card0-DP-1)
        try:
            if "amdgpu" not in os.path.realpath(os.path.join(dev, "driver")):
                continue
            cards.append((os.path.basename(os.path.realpath(dev)), dev))
        except OSError:
            continue
    for _, dev in sorted(cards):
        try:
            nodes.append(int(open(os.path.join(dev, "numa_node")).read().strip()))
        except (OSError, ValueError):
            nodes.append(-1)
    return nodes


def visible_gpu_order(n_physical, env=None):
    """Physical indices of the GPUs this process will see, in the order HIP numbers them: ROCR_VISIBLE_DEVICES filters (and reorders) first, then
    HIP_VISIBLE_DEVICES (or CUDA_VISIBLE_DEVICES) picks among what is left.  Entries that are not plain indices (UUIDs) leave the mapping unknown: None."""
    env = os.environ if env is None else env
    order = list(range(n_physical))
    for name in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES" if "HIP_VISIBLE_DEVICES" in env else "CUDA_VISIBLE_DEVICES"):
        val = env.get(name)
        if val is None or val.strip() == "":
            continue
        picked = []
        for part in val.split(","):
            part = part.strip()
            if not part.lstrip("-").isdigit():
                return None
            k = int(part)
            if k < 0 or k >= len(order):
                break                                             # (the runtime stops at the first invalid entry)
            picked.append(order[k])
        order = picked
    return order


def plan_affinity(local_rank, world, nodes, node_cpus, allowed):
    """The CPU set of rank `local_rank`: the allowed cores of its GPU's NUMA node, dealt evenly to the ranks whose GPUs share that node
    (rank r takes every k-th core from its position among them).  None = leave the affinity alone (unknown topology, or too few cores)."""
    if local_rank >= len(nodes) or nodes[local_rank] < 0:
        return None
    node = nodes[local_rank]
    cores = sorted(node_cpus.get(node, set()) & allowed)
    sharers = [r for r in range(min(world, len(nodes))) if nodes[r] == node]
    if local_rank not in sharers or len(cores) < 2 * len(sharers):
        return None
    k = sharers.index(local_rank)
    per = len(cores) // len(sharers)
    return set(cores[k * per:(k + 1) * per])


def pin_to_gpu_node(local_rank, world):
    """Pin this process before it initialises a GPU; returns a short description for the JSON line."""
    if os.environ.get("BENCH_NO_PIN"):
        return "unchanged (BENCH_NO_PIN)"
    try:
        allowed = os.sched_getaffinity(0)
        nodes = gpu_numa_nodes()
        # a launcher that masks or reorders the devices (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES): rank r's GPU is the r-th VISIBLE one
        order = visible_gpu_order(len(nodes))
        if order is None:
            return f"unchanged ({len(allowed)} cores; the visible-devices mask is not a list of indices)"
        nodes = [nodes[k] for k in order]
        import glob
        node_cpus = {}
        for nd in glob.glob("/sys/devices/system/node/node[0-9]*"):
            try:
                node_cpus[int(os.path.basename(nd)[4:])] = _parse_cpulist(open(os.path.join(nd, "cpulist")).read())
            except (OSError, ValueError):
                pass
        want = plan_affinity(local_rank, world, nodes, node_cpus, allowed)
        if not want:
            return f"unchanged ({len(allowed)} cores; GPU NUMA nodes {nodes or 'unknown'})"
        os.sched_setaffinity(0, want)
        lo, hi = min(want), max(want)
        return f"{len(want)} cores {lo}-{hi} of NUMA node {nodes[local_rank]} (GPU {local_rank})"
    except Exception as e:                                        # never fail a bench run over placement
        return f"unchanged ({type(e).__name__}: {e})"


def progress(msg):
    """A rank's milestone on stderr, `[bench rank R +T.Ts] msg`: what a self-launching parent shows for every rank when the run exceeds its budget."""
    r = os.environ.get("RANK", "0")
    print(f"[bench rank {r} +{time.perf_counter() - _T_START:.1f}s] {msg}", file=sys.stderr, flush=True)


_T_START = time.perf_counter()


def run_ranks(cmd, env, timeout_s, out=sys.stderr):
    """Run the rank launcher `cmd` as a FRESH child in a process group of its own (never a re-exec: nothing here has touched a GPU, and nothing that has
    may exec), forward its stderr line by line while remembering what every rank logged last, and give it `timeout_s` seconds of wall clock (0: no limit).
    Beyond the budget the whole process group is killed (SIGTERM, then SIGKILL), every rank's last lines are printed and the exit code is 124 -- a hung
    RCCL initialisation on a node this repository has never seen then costs the budget, not the caller's whole lease.  Returns (rc, stdout bytes)."""
    import signal
    import subprocess
    import threading
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
    last = {}                                                   # rank (or "-") -> its last few stderr lines
    chunks = []

    def pump_err():
        for raw in iter(proc.stderr.readline, b""):
            t = raw.decode(errors="replace").rstrip("\n")
            print(t, file=out, flush=True)
            key = "-"
            if t.startswith("[bench rank "):
                key = t[len("[bench rank "):].split(" ", 1)[0].rstrip("]")
            elif t.startswith("[rank") or t.startswith("[default"):      # torch.distributed.run's own prefixes
                key = t[1:].split("]", 1)[0]
            last.setdefault(key, []).append(t)
            del last[key][:-3]

    def pump_out():
        for raw in iter(lambda: proc.stdout.read(65536), b""):
            chunks.append(raw)

    threads = [threading.Thread(target=pump_err, daemon=True), threading.Thread(target=pump_out, daemon=True)]
    for th in threads:
        th.start()
    timed_out = False
    try:
        proc.wait(timeout=timeout_s if timeout_s and timeout_s > 0 else None)
    except subprocess.TimeoutExpired:
        timed_out = True
        for sig, grace in ((signal.SIGTERM, 5.0), (signal.SIGKILL, 5.0)):
            try:
                os.killpg(proc.pid, sig)                        # (start_new_session: the child's pid is its process group's id -- exactly the ranks, nothing else)
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=grace)
                break
            except subprocess.TimeoutExpired:
                continue
    for th in threads:
        th.join(timeout=5.0)
    if timed_out:
        print(f"bench.py: the ranks did not finish within --launch-timeout {timeout_s:.0f} s; their process group was killed.  Last lines per rank:", file=out)
        for key in sorted(last):
            for t in last[key]:
                print(f"    rank {key}: {t}", file=out)
        if not last:
            print("    (no rank had logged anything)", file=out)
        out.flush()
        return 124, b"".join(chunks)
    return proc.returncode, b"".join(chunks)


def self_launch(args, argv):
    """Start the ranks as a fresh child (nothing in THIS process has touched a GPU, torch is not even imported) and
    hand on its exit code.  The child's rank 0 writes the JSON line to the stdout it inherits.  The child runs in a process group of its own under a
    wall-clock budget (--launch-timeout, default 600 s): see run_ranks."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    sys.stdout.flush()
    # the ranks keep their stdout clean themselves (guard_stdout); should anything still reach it, only the last JSON line is passed on
    rc, stdout_bytes = run_ranks(cmd, env, args.launch_timeout)
    line = None
    for raw in stdout_bytes.decode(errors="replace").splitlines():
        t = raw.strip()
        if t.startswith("{") and t.endswith("}"):
            try:
                json.loads(t)
                line = t
                continue
            except ValueError:
                pass
        if t:
            print(raw, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    return rc


def cpu_baseline(a, dt, fibre, sample_steps):
    """The oracle (NumPy restatement of the reference, single thread like the reference) timed on
    a bounded sample of the SAME workload: `sample_steps` SSFM steps of the 2^20 x 2 field."""
    import numpy as np
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


def secondary_big(dt, fibre, device, log2n=24, steps=20):
    """The largest field the library takes (round 6: 2^24 samples x 2 polarisations, a "split plan" -- DESIGN.md 5b), reported beside the headline: 20 of C2's steps on a
    synthetic complex64 field (band-limited noise: the content does not matter to the timing), with the one property that can be checked without an oracle run of minutes --
    the energy follows the float32 attenuation factor (every other operator of a step is unitary)."""
    import numpy as np
    from opticomlib_amd import _lib, devices
    n = 1 << log2n
    rng = np.random.default_rng(log2n)
    a = np.empty((N_POL, n), dtype=np.complex64)
    for r in range(N_POL):                                   # (smoothed white noise, 1 mW per polarisation; built row by row: 2^24 samples per call)
        w = (rng.standard_normal(n, dtype=np.float32) + 1j * rng.standard_normal(n, dtype=np.float32)).astype(np.complex64)
        w = w + np.roll(w, 1) + np.roll(w, 2) + np.roll(w, 3)
        a[r] = w * np.float32(np.sqrt(1e-3 / np.mean(np.abs(w[: 1 << 16]) ** 2)))
    hs = np.full(steps, H_KM, np.float32)
    p = _lib.Plan(n, N_POL, _lib.C64, device=device)
    try:
        p.set_linear_operator(devices.linear_operator(n, dt, fibre["alpha"], fibre["beta_2"], fibre["beta_3"], _lib.C64))
        p.set_field(a)
        p.propagate_fixed(fibre["gamma"], hs)
        p.synchronize()
        y = p.get_field()
        e_in = np.sum(np.abs(a.astype(np.complex128)) ** 2, axis=-1)
        e_out = np.sum(np.abs(y.astype(np.complex128)) ** 2, axis=-1)
        att = float(np.exp(np.complex64(-np.float32(fibre["alpha"] / 4.343) / 2) * np.float32(H_KM)).real)
        energy_err = float(np.max(np.abs(e_out / e_in / att ** (2 * steps) - 1.0)))
        reps = 3
        t = time.perf_counter()
        for _ in range(reps):
            p.propagate_fixed(fibre["gamma"], hs)
        p.synchronize()
        el = (time.perf_counter() - t) / reps
        info = p.last_run_info()
    finally:
        p.close()
    rate = n * steps / el
    return {"workload": f"2^{log2n}-sample dual-pol field (256 MiB), complex64, {steps} of C2's steps (h = {H_KM} km), synthetic band-limited noise", "dtype": "c64", "engine": info["engine"],
            "value": rate, "unit": "sample*steps/s", "us_per_ssfm_step": el / steps * 1e6, "step_frac": 2 * N_POL * 8 * rate / (HBM_PEAK_GBS * 1e9),
            "energy_against_attenuation": energy_err,
            "note": "rows of more than 2^22 samples run as split plans: four passes per step (DESIGN.md 5b); algorithmic bytes 32 B per sample*step; not part of `value`"}


def secondary_filter(a_c2, device):
    """The zero-phase Bessel filter of LPF / BPF on the headline's field, device-resident: a 4th-order BPF of the 2^20 x 2 complex128 field, kernels
    per call by the library's own HIP events, with the check against SciPy on one row.  x read once + y written once = 32 B per complex sample: its roofline."""
    import numpy as np
    from scipy import signal as sg
    from opticomlib_amd import _lib
    n = a_c2.shape[-1]
    from opticomlib_amd import workloads
    sos = sg.bessel(4, 30e9, "low", fs=workloads.BENCH_GV["sps"] * workloads.BENCH_GV["R"], norm="mag", output="sos")
    zi = sg.sosfilt_zi(sos)
    x = a_c2.astype(np.complex128)
    p, q = _lib.Plan(n, N_POL, _lib.C128, device=device), _lib.Plan(n, N_POL, _lib.C128, device=device)
    try:
        p.set_field(x)
        p.synchronize()
        for _ in range(20):
            _lib.sosfiltfilt_device(sos, zi, p.field_device_ptr, q.field_device_ptr, n, N_POL, True, device)
        err = float(np.abs(q.get_field()[0] - sg.sosfiltfilt(sos, x[0])).max() / np.abs(x[0]).max())
        t = 0.0
        reps = 200
        for _ in range(reps):
            _lib.sosfiltfilt_device(sos, zi, p.field_device_ptr, q.field_device_ptr, n, N_POL, True, device)
            t += _lib.sosfiltfilt_last_ms()
        us = t / reps * 1e3
        launches = _lib.sosfiltfilt_last_launches()
    finally:
        p.close()
        q.close()
    nbytes = 2 * 16 * n * N_POL
    return {"workload": "BPF (4th-order Bessel, zero phase) of the 2^20-sample dual-pol field, complex128, device-resident, 200 calls", "dtype": "f64", "kernel_us_per_call": us,
            "launches_per_call": launches, "algorithmic_bytes_per_call": nbytes, "achieved_GBs": nbytes / us / 1e3, "frac": nbytes / us / 1e3 / HBM_PEAK_GBS,
            "against_scipy": err, "note": "kernels by the library's HIP events around each call; not part of `value`"}


def secondary_any_length(dt, fibre):
    """A length that is NOT a power of two, at the reference's own size: the PRBS-16 word at 16 samples per bit, (2^16 - 1) * 16 = 1 048 560 samples x 2.
    Per step: the slope between a 100-step and a 400-step FIBER() call (host arrays in and out: the call's fixed costs cancel); beside it the ratio to the headline's step."""
    import numpy as np
    import opticomlib_amd as oa
    from opticomlib_amd import workloads
    from opticomlib_amd.typing import gv, optical_signal
    gv(**workloads.BENCH_GV)
    n = ((1 << 16) - 1) * 16
    a = workloads.qpsk_field(n, seed=1616, n_pol=N_POL, power_w=2e-3)
    x = optical_signal(a)
    t = {}
    for steps in (100, 400):
        kw = dict(length=steps * H_KM, h=H_KM, **fibre)
        if steps == 100:
            y = oa.FIBER(x, **kw).signal
            e_in = np.sum(np.abs(a.astype(np.complex128)) ** 2, axis=-1)
            e_out = np.sum(np.abs(y.astype(np.complex128)) ** 2, axis=-1)
            att = float(np.exp(np.complex64(-np.float32(fibre["alpha"] / 4.343) / 2) * np.float32(H_KM)).real)
            energy_err = float(np.max(np.abs(e_out / e_in / att ** (2 * steps) - 1.0)))
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            oa.FIBER(x, **kw)
            best = min(best, time.perf_counter() - t0)
        t[steps] = best
    oa.devices.release_plans()
    us = (t[400] - t[100]) / 300 * 1e6
    return {"workload": f"{n}-sample dual-pol field ((2^16 - 1) * 16: not a power of two), complex64 caller, C2's steps (h = {H_KM} km)", "dtype": "c64 between f64 passes",
            "us_per_ssfm_step": us, "value": n / (us * 1e-6), "unit": "sample*steps/s", "fixed_ms_per_call": (4 * t[100] - t[400]) / 3 * 1e3,
            "energy_against_attenuation": energy_err,
            "note": "chirp-z line of 2^21 points, four passes per step on two lanes; slope between a 100-step and a 400-step FIBER() call; not part of `value`"}


def _manycore_worker(job):
    seed, dt, fibre, steps = job
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    import numpy as np
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


def committed_profile_figures(kernels, rows_per_launch):
    """Figures that cannot be taken inside a bench run (rocprofv3 is a separate process; PMC counters need their own
    passes): read from the newest committed summaries under profiles/ and reported under `from_profiles`, apart from
    what this run measured."""
    out = {}
    for stats_name in ("r06_final_kernel_stats.csv", "r05_final_kernel_stats.csv", "r04_final_kernel_stats.csv", "r03_final_kernel_stats.csv", "r02_final_kernel_stats.csv"):
        stats = os.path.join(ROOT, "profiles", stats_name)
        if not os.path.exists(stats):
            continue
        try:
            import csv
            acc = {}
            for row in csv.DictReader(open(stats)):
                for k in kernels:
                    if k + "<float" in row["Name"]:          # the C2 kernels (the file may also hold C1's complex128 ones)
                        c, tns = acc.get(k, (0, 0.0))
                        acc[k] = (c + int(row["Calls"]), tns + float(row["TotalDurationNs"]))
            out["rocprof_kernel_us"] = {k: tns / c / 1e3 for k, (c, tns) in acc.items() if c}
            out["rocprof_source"] = f"profiles/{stats_name} (rocprofv3 --kernel-trace --stats of this command: kernel begin->end, no launch gap)"
        except Exception:
            pass
        break
    for pmc_name in ("r06_final_pmc_traffic.json", "r05_final_pmc_traffic.json", "r04_final_pmc_traffic.json", "r03_final_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"):
        pmc = os.path.join(ROOT, "profiles", pmc_name)
        if not os.path.exists(pmc):
            continue
        try:
            t = json.load(open(pmc))
            # (the PMC passes were taken on one-row launches; a launch over more rows moves that many times the field bytes)
            per_row = {k: t.get(k, {}).get("bytes_per_launch") for k in kernels}
            out["traffic_per_kernel"] = {k: (v * rows_per_launch if v else None) for k, v in per_row.items()}
            out["traffic_source"] = f"profiles/{pmc_name} (FETCH_SIZE x 2 + WRITE_SIZE per one-row launch, x rows_per_launch): " + str(t.get("_source"))
        except Exception:
            pass
        break
    return out


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if "RANK" not in os.environ and (args.gpus > 1 or args.workload in ("c3", "c4")):
        sys.exit(self_launch(args, argv))

    guard_stdout()                                    # before any library can print on it
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    cpu_affinity = pin_to_gpu_node(0 if os.environ.get("BENCH_SAME_GPU") else local_rank, int(os.environ.get("WORLD_SIZE", "1")))

    import numpy as np
    import torch
    import torch.distributed as dist

    if os.environ.get("BENCH_SAME_GPU"):             # diagnostics only: every rank on GPU 0 (needs BENCH_DIST_BACKEND=gloo: RCCL wants one GPU per rank)
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} is running with WORLD_SIZE={world}: start it as `python bench.py --gpus {args.gpus}` "
                         f"(it launches its own ranks) or under torch.distributed.run with --nproc-per-node {args.gpus}")
    visible = torch.cuda.device_count()               # (counting devices does not initialise the GPU)
    if visible < (1 if os.environ.get("BENCH_SAME_GPU") else world):
        raise SystemExit(f"bench.py --gpus {world} needs {world} GPUs on this node: {visible} visible "
                         f"(rank {rank}: torch.cuda.device_count() = {visible})")
    distributed = "RANK" in os.environ          # under torch.distributed.run, also with one rank
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")          # diagnostics only: gloo | none
    if backend == "none":
        distributed = False
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        progress(f"init_process_group({backend}) of {world} rank(s), GPU {local_rank} ...")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        progress("process group up")

    from opticomlib_amd import _lib, devices, workloads
    from opticomlib_amd import dist as od

    n = 1 << LOG2N
    fibre = dict(workloads.SMF)
    dt = 1.0 / (workloads.BENCH_GV["sps"] * workloads.BENCH_GV["R"])
    workload = "c2" if args.workload == "auto" else args.workload
    lin = lambda sign, prec=_lib.C64: devices.linear_operator(n, dt, sign * fibre["alpha"], sign * fibre["beta_2"], sign * fibre["beta_3"], prec)  # noqa: E731

    plans = []
    a_c2 = None
    if workload in ("c2", "c3"):
        if workload == "c3":
            total_fields, scaling = C3_FIELDS, "strong"
            units = od.shard(C3_FIELDS, rank, world)                  # round-robin: unit i on rank i % world
            seeds = [3000 + u for u in units]
        else:
            total_fields, scaling = world, "weak"
            units = [rank]
            seeds = [2024] if world == 1 else [3000 + rank]           # N = 1: the C2 seed; N > 1: the WDM channels of C3
        fields_here = len(units)
        steps_per_field = SSFM_STEPS
        hs, _ = devices.step_schedule(LENGTH_KM, H_KM, _lib.C64)
        assert hs.size == SSFM_STEPS
        plan = None
        if fields_here:
            a = np.stack([workloads.qpsk_field(n, seed=s, n_pol=N_POL) for s in seeds])
            a_c2 = a[0]
            plan = _lib.Plan(n, N_POL * fields_here, _lib.C64, device=local_rank)
            plans.append(plan)
            plan.set_linear_operator(lin(1.0))
            x_dev = torch.from_numpy(np.ascontiguousarray(a.astype(np.complex64))).cuda()       # resident input

        def one_step():
            if plan is None:
                return
            plan.set_field_device(x_dev.data_ptr())          # D2D restore on the plan's stream
            plan.propagate_fixed(fibre["gamma"], hs)         # 1 + 2*1000 launches per lane, asynchronous

        result_ptr = (lambda: plan.field_device_ptr if plan is not None else 0)
        result_owner = plan
    else:
        # C4: this rank's realisations are generated in GPU memory from their LFSR seeds before the timed region
        total_fields, scaling = C4_SEEDS, "strong"
        units = od.shard(C4_SEEDS, rank, world)
        fields_here = len(units)
        steps_per_field = 200
        hs, _ = devices.step_schedule(C4_LENGTH_KM, C4_H_KM, _lib.C64)
        assert hs.size == 100
        per = min(C4_RESIDENT, max(fields_here, 1))
        assert fields_here % per == 0, "C4: the realisations of a rank must fill whole resident blocks"
        x_all = _lib.DeviceArray((max(fields_here, 1), N_POL, n), np.complex64, local_rank)
        y_all = _lib.DeviceArray((max(fields_here, 1), N_POL, n), np.complex64, local_rank)
        fb = N_POL * n * 8
        for k, u in enumerate(units):
            f = workloads.prbs_field_device(n, seed=1 + u, power_w=1e-3, device=local_rank)
            _lib._check(_lib.load().ssfm_device_copy(local_rank, _lib._VP(x_all.ptr + k * fb), _lib._VP(f.ptr), fb, 2), "ssfm_device_copy")
        devices.release_plans()                              # (the generator's complex128 plan)
        plan_f = plan_b = None
        if fields_here:
            plan_f = _lib.Plan(n, N_POL * per, _lib.C64, device=local_rank)      # FIBER: its operator stays resident
            plan_b = _lib.Plan(n, N_POL * per, _lib.C64, device=local_rank)      # DBP = FIBER with every parameter negated (devices.py:1280-1283)
            plans += [plan_f, plan_b]
            plan_f.set_linear_operator(lin(1.0))
            plan_b.set_linear_operator(lin(-1.0))
        plan = plan_f

        def one_step():
            for b0 in range(0, fields_here, per):
                plan_f.set_field_device(x_all.ptr + b0 * fb)
                plan_f.propagate_fixed(fibre["gamma"], hs)
                plan_f.synchronize()                                             # plan_b's stream reads plan_f's buffer
                plan_b.set_field_device(plan_f.field_device_ptr)
                plan_b.propagate_fixed(-fibre["gamma"], hs)
                plan_b.get_field_device(y_all.ptr + b0 * fb)
                plan_b.synchronize()                                             # before plan_f's buffer is overwritten

        result_ptr = (lambda: y_all.ptr)
        result_owner = y_all
    torch.cuda.synchronize()

    def fence():
        for p in plans:
            p.synchronize()                              # the plans' own (non-blocking, high-priority) streams
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # plan set-up, not part of the protocol's warm-up: first touch of every buffer and table
    progress(f"plans made ({workload}, {fields_here} field(s) here); first pass ...")
    one_step()
    fence()
    for _ in range(args.warmup):
        one_step()
    fence()
    progress(f"warm-up done; timing {args.steps} step(s) ...")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    fence()
    elapsed = time.perf_counter() - t0
    elapsed_here = elapsed
    progress(f"timed region done: {elapsed / max(args.steps, 1) * 1e3:.2f} ms per bench step on this rank")
    ranks_seen, per_rank_s = 1, [elapsed_here]
    if distributed:
        dev_t = "cuda" if backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev_t)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank answers: a sum of ones over the group, and each rank's OWN time beside the maximum -- a slow or missing rank shows in the one JSON line
        ones = torch.ones(1, dtype=torch.float64, device=dev_t)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        ranks_seen = int(round(float(ones.item())))
        mine = torch.tensor([elapsed_here], dtype=torch.float64, device=dev_t)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_s = [float(v.item()) for v in every]

    ms_dev, launches = plan.last_propagate_ms() if plan is not None else (0.0, 0)
    run_info = plan.last_run_info() if plan is not None else None          # which engine ran, whether it fell back, whether the lanes share a queue
    if workload == "c4":
        out = y_all.to_host()[:fields_here]
    else:
        out = plan.get_field().reshape(fields_here, N_POL, n) if plan is not None else np.zeros((0, N_POL, n), np.complex64)
    powers = [float(np.mean(np.abs(o.astype(np.complex128)) ** 2)) for o in out]

    # the only "exchange" of this path: the gather of the results at the end, in GPU memory (RCCL over xGMI)
    gather_ms = None
    collectives = None
    checks = powers
    if distributed and backend == "nccl":
        def gather():
            return od.gather_device(result_ptr(), fields_here, (N_POL, n), np.complex64, total_fields, local_rank, to_all=False, owner=result_owner)
        got = gather()                                         # warm-up: communicator set-up
        reps = 3
        fence()
        tg = time.perf_counter()
        for _ in range(reps):
            got = gather()
        fence()
        gather_ms = (time.perf_counter() - tg) / reps * 1e3
        progress(f"gather done: {gather_ms:.2f} ms")
        collectives = dict(od.COLLECTIVES)
        if rank == 0:
            g = got.to_host()
            checks = [float(np.mean(np.abs(g[u].astype(np.complex128)) ** 2)) for u in range(total_fields)]
            mine = od.shard(total_fields, 0, world)
            for k, u in enumerate(mine):                                                      # the gathered block holds this rank's own results in place
                assert np.array_equal(g[u], out[k]), "gathered field differs from the local result"
    elif distributed:
        gl = [torch.zeros(max(1, -(-total_fields // world)), dtype=torch.float64) for _ in range(world)]
        mine_t = torch.zeros_like(gl[0])
        mine_t[: len(powers)] = torch.tensor(powers, dtype=torch.float64)
        dist.all_gather(gl, mine_t)
        checks = [float(v) for t in gl for v in t.tolist()]

    value = total_fields * n * steps_per_field * args.steps / elapsed                   # whole job: every field of every rank

    roofline = None
    cpu = None
    cpu_many = None
    other = None
    big = None
    extra = {}
    if rank == 0:
        per_gpu_rate = value * fields_here / max(total_fields, 1)
        roofline = {"bound": "hbm", "kernel": None, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                    "step_frac": 2 * N_POL * 8 * per_gpu_rate / (HBM_PEAK_GBS * 1e9),
                    "step_frac_note": "whole step on this GPU: 32 B per sample*step (dual-pol complex64 field read once + written once) x its sample*steps/s / peak"}
        if not args.no_profile_pass and workload != "c4" and plan is not None:
            lanes = plan.lanes
            rows_per_launch = N_POL * fields_here // lanes       # a launch covers one lane's rows
            b_alg_launch = 2 * rows_per_launch * 8 * n           # its rows read once + written once

            def timed_pass(mode):
                plan.set_profiling(mode)
                one_step()
                plan.synchronize()
                kt = plan.kernel_times()
                plan.set_profiling(0)
                return kt
            # (1) The launch duration that prices the roofline comes from the TIMED REGION itself, not from an extra pass: the plan brackets every
            # propagate call with two HIP events on its own stream (ssfm_last_propagate_ms); a lane is a strictly serial chain of launches, so
            # device time / launches per lane = the average launch PERIOD on that stream = kernel + the dependent-launch gap behind it.
            launches_per_lane = launches / max(lanes, 1)
            period_us = ms_dev * 1e3 / max(launches_per_lane, 1)
            # (2) Two extra passes after the timed region only tell the two kernels apart: one launch of each per 16 bracketed by two events of its
            # own (the brackets add the markers' own cost, so only the RATIO is used), and one event per 64 launches as a cross-check of (1).
            sparse = timed_pass(2)
            sampled = timed_pass(3)
            pooled_us = sum(v[1] for v in sparse.values()) / max(sum(v[0] for v in sparse.values()), 1) * 1e3
            bracketed_us = {k: (v[1] / v[0] * 1e3 if v[0] else None) for k, v in sampled.items()}
            ok = all(v is not None for v in bracketed_us.values())
            mean_b = sum(bracketed_us.values()) / len(bracketed_us) if ok else None
            launch_us = {k: (period_us * bracketed_us[k] / mean_b if ok else period_us) for k in bracketed_us}
            dom = max(launch_us, key=lambda k: launch_us[k] or 0.0)
            avg_us = launch_us[dom]
            achieved = b_alg_launch / (avg_us * 1e-6) / 1e9
            gap_us = 1.35                                  # dependent-launch gap on a stream (profiles/r04_c2_stamps_and_trace.txt: 1.2-1.4 us; MI355X_MICROARCH.md "boundary")
            prof = committed_profile_figures(list(launch_us), rows_per_launch)
            tpk = prof.get("traffic_per_kernel") or {}
            if tpk and all(tpk.get(k) for k in launch_us):
                # the step priced by the bytes it REALLY moves (PMC, committed profile): every lane runs one launch of each kernel per step
                step_bytes = lanes * sum(tpk[k] for k in launch_us)
                roofline["step_traffic"] = {"bytes_per_step": step_bytes, "rate_GBs": step_bytes / (period_us * 1e-6 * len(launch_us)) / 1e9,
                                            "frac_of_peak": step_bytes / (period_us * 1e-6 * len(launch_us)) / 1e9 / HBM_PEAK_GBS,
                                            "note": "PMC bytes per launch (read_from_profiles) x lanes x kernels per step / the measured step time (launch period x kernels per step)"}
            roofline.update({
                "kernel": dom, "achieved": achieved, "frac": achieved / HBM_PEAK_GBS,
                "traffic": (prof.get("traffic_per_kernel") or {}).get(dom),
                "avg_launch_us": avg_us,
                "launch_us": launch_us,
                "launch_period_us": period_us,
                "kernel_us_without_gap": {k: v - gap_us for k, v in launch_us.items()},
                "assumed_gap_us": gap_us,
                "launches_sampled": {k: v[0] for k, v in sampled.items()},
                "bracketed_launch_us": bracketed_us,
                "pooled_launch_us": pooled_us,
                "algorithmic_bytes_per_launch": b_alg_launch,
                "lanes": lanes, "rows_per_launch": rows_per_launch,
                "measured_in_this_run": ["achieved", "frac", "avg_launch_us", "launch_us", "launch_period_us", "launches_sampled", "bracketed_launch_us",
                                         "pooled_launch_us", "step_frac"],
                "read_from_profiles": ["traffic", "from_profiles"],
                "from_profiles": prof,
                "note": "avg_launch_us = the dominant kernel's average launch period on its stream in the TIMED region: HIP events of the plan around the "
                        "propagate call (device_ms_last_propagate) / launches per lane, split between k_time and k_freq in the ratio of their bracketed "
                        "times (second extra pass).  A period = kernel + the dependent-launch gap behind it, so `achieved` is a lower bound of the kernel's "
                        "own rate (kernel_us_without_gap subtracts the stated gap).  With lanes > 1 each lane is such a chain and the chains run side by "
                        "side: frac equals step_frac when both kernels take the same time.  rocprofv3's begin->end averages (from_profiles) are taken "
                        "under the profiler, which slows the run by a few per cent",
            })
        if world == 1 and workload == "c2" and not args.no_secondary:
            other = secondary_c1(a_c2, dt, fibre, local_rank)
            try:
                big = None if args.no_big_field else secondary_big(dt, fibre, local_rank)
            except Exception as e:                               # (reported, never fatal: the headline line stands on its own)
                big = {"error": f"{type(e).__name__}: {e}"}
            extra = {}
            for name, fn in (("secondary_filter", lambda: secondary_filter(a_c2, local_rank)), ("secondary_any_length", lambda: secondary_any_length(dt, fibre))):
                if args.no_big_field:
                    continue
                try:
                    extra[name] = fn()
                except Exception as e:
                    extra[name] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and args.cpu_steps > 0:
            if a_c2 is None:
                a_c2 = workloads.qpsk_field(n, seed=2024, n_pol=N_POL)
            cpu = cpu_baseline(a_c2, dt, fibre, args.cpu_steps)
        if world == 1 and args.cpu_manycore > 0:
            cpu_many = cpu_baseline_manycore(dt, fibre, args.cpu_manycore, max(8, args.cpu_steps // 4))

    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    names = {
        "c2": ("C2: 2^20-sample dual-pol optical_signal, FIBER(length=125, h=0.125) = 1000 SSFM steps, complex64" if world == 1 else
               f"C2 on every GPU: one 2^20-sample dual-pol field per rank (WDM channels, seeds 3000 + rank; at 8 GPUs = configuration C3), "
               f"FIBER(length=125, h=0.125) = 1000 SSFM steps each, complex64"),
        "c3": (f"C3: {C3_FIELDS} independent 2^20-sample dual-pol fields (WDM channels, seeds 3000..), FIBER(length=125, h=0.125) = 1000 SSFM steps each, "
               f"complex64, unit i on rank i % {world}, a rank's fields batched in one plan"),
        "c4": (f"C4: {C4_SEEDS} Monte-Carlo PRBS realisations (LFSR seeds 1..{C4_SEEDS}, generated on the device), 2^20-sample dual-pol, FIBER(100 x 1 km) + "
               f"DBP(100 x 1 km) back to back in GPU memory, complex64, unit i on rank i % {world}, {C4_RESIDENT} resident per plan"),
    }
    emit_json_line({
        "metric": "SSFM sample*steps/sec, 2^20-sample dual-pol fiber",
        "value": value,
        "unit": "sample*steps/s",
        "n_gpus": world,
        "ranks_seen": ranks_seen,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "c64",
        "data": "synthetic",
        "config": {
            "workload": names[workload],
            "n_samples": n, "n_pol": N_POL, "ssfm_steps_per_field": steps_per_field,
            "fields_total": total_fields, "fields_per_gpu": fields_here, "parallelism": f"independent-fields x{world}",
        },
        **({"gather_ms": gather_ms, "gather_bytes": total_fields * N_POL * n * 8, "collectives_issued": collectives,
            "gather_note": "all propagated fields to rank 0 in GPU memory, one RCCL gather on the plans' field buffers; after the timed region, not in `value`"}
           if gather_ms is not None else {}),
        "device_ms_last_propagate": ms_dev,
        "launches_per_propagate": launches,
        "engine": run_info,
        "us_per_ssfm_step": elapsed / args.steps / steps_per_field * 1e6 / max(fields_here, 1),
        "us_per_ssfm_step_note": "wall time of a bench step / SSFM steps per field / fields on this GPU (per field-step)",
        "us_per_ssfm_step_by_rank": [v / args.steps / steps_per_field * 1e6 / max(len(od.shard(total_fields, r, world)), 1) for r, v in enumerate(per_rank_s)],
        "us_per_ssfm_step_by_rank_note": "every rank's OWN wall time of the timed region per field-step (`value` uses the maximum over ranks); `ranks_seen` = an all-reduce of ones",
        "output_power_W_per_field": checks,
        "roofline": roofline,
        "cpu_baseline": cpu,
        **({"cpu_baseline_manycore": cpu_many} if cpu_many else {}),
        **({"secondary": other} if other else {}),
        **({"secondary_big_field": big} if big else {}),
        **extra,
        "cpu_affinity": cpu_affinity,
    })


if __name__ == "__main__":
    main()
