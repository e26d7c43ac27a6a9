"""Round 6: what a z-resolved capture costs an ADAPTIVE run (2^20 x 2 complex64, the long run of tools/attic/adapt_long.py: 652 steps): the plain run, the capture
that keeps the run's engine (ssfm_adaptive_set_capture) at every = 100 / 50 / 20 / 10, and the every-step capture of ssfm_adaptive_run(snapshots) over the first 60 steps.
Wall clock until the caller's buffers are valid; best of 3, interleaved.   python tools/adaptive_capture_time.py > gpurun_out/r06_adaptive_capture.txt"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from opticomlib_amd import _lib, devices, workloads

n = 1 << 20; dt = 1.0 / (16 * 32e9)
a = workloads.qpsk_field(n, seed=1, power_w=10e-3).astype(np.complex64)
p = _lib.Plan(n, 2, _lib.C64); p.set_linear_operator(devices.linear_operator(n, dt, 0.2, -21.7, 0.13))
res = {}
dest = {every: _lib.host_empty((700 // every + 1, 2, n), np.complex64, limit=8 << 30) for every in (100, 50, 20, 10)}      # (made once: page-locking a GiB takes longer than the run)
for rnd in range(3):
    p.set_field(a); p.synchronize()
    t = time.perf_counter(); s, z, _ = p.propagate_adaptive(1.3, 80.0, 0.002, False); el = time.perf_counter() - t
    res.setdefault("plain", []).append(el / s * 1e6)
    eng = p.last_run_info()["engine"]
    for every in (100, 50, 20, 10):
        p.set_field(a); p.synchronize()
        t = time.perf_counter(); s2, z2, taken, fields = p.propagate_adaptive_capture(1.3, 80.0, 0.002, every=every, capacity=700 // every + 1, fields=dest[every]); el = time.perf_counter() - t
        assert s2 == s and len(taken) == s // every, (s2, s, len(taken))
        res.setdefault(f"every {every:3d} ({len(taken)} snapshots of 16 MiB)", []).append(el / s * 1e6)
    p.set_field(a); p.synchronize()
    lib = _lib.load(); st, dn = _lib._I64(0), _lib._I(0)
    blk = _lib.host_empty((60, 2, n), np.complex64, limit=8 << 30)
    t = time.perf_counter()
    _lib._check(lib.ssfm_adaptive_begin(p._h, 1.3, 80.0, 0.002, 0, 1 << 16, 1), "begin")
    _lib._check(lib.ssfm_adaptive_run(p._h, 60, blk.ctypes.data_as(_lib.C.c_void_p), _lib.C.byref(st), _lib.C.byref(dn)), "run")
    el = time.perf_counter() - t
    zb = np.zeros(st.value + 1); _lib._check(lib.ssfm_adaptive_finish(p._h, _lib.C.byref(st), zb.ctypes.data_as(_lib.C.POINTER(_lib._D))), "finish")
    res.setdefault("every step, the host waits per step (round 4's form, first 60 steps)", []).append(el / 60 * 1e6)
print(f"2^20 x 2 complex64, adaptive, {s} steps, engine of the plain run: {eng}")
base = min(res["plain"])
for k, v in res.items():
    print(f"  {k:70s} {min(v):7.2f} us per step  ({(min(v) / base - 1) * 100:+5.1f} %)   [{', '.join(f'{x:.2f}' for x in v)}]")
p.close()
