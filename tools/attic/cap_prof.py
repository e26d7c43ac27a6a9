"""Where does a capture run's time go?  Plain and capture runs of 1000 steps in turn: wall time per step and the device time between the run's own events."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import opticomlib_amd as oa
from opticomlib_amd import _lib, workloads
from opticomlib_amd.typing import gv
gv(**workloads.BENCH_GV)
n = 1 << 20
a = workloads.qpsk_field(n, seed=2024).astype(np.complex64)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
hs = np.full(steps, 0.125, np.float32)
p = _lib.Plan(n, 2, _lib.C64)
p.set_linear_operator(oa.devices.linear_operator(n, gv.dt, 0.2, -21.7, 0.13))
lib = _lib.load()
import ctypes as C
raw = _lib.host_empty((steps + 1, 2, 2), np.float64)
snaps = _lib.host_empty((steps // 50 + 2, 2, n), np.complex64, limit=8 << 30)
for rep in range(3):
    for mode in os.environ.get("CAP_MODES", "plain,scal,every100,plain").split(","):
        p.set_field(a); p.synchronize()
        t0 = time.perf_counter()
        if mode == "plain":
            p.propagate_fixed(1.3, hs)
        else:
            cap = _lib.Capture(0, None, None)
            if mode == "scal": cap.scalars = raw.ctypes.data
            else: cap.every, cap.fields = int(mode[5:]), snaps.ctypes.data
            _lib._check(lib.ssfm_propagate_fixed_capture(p._h, 1.3, hs.ctypes.data, hs.size, C.byref(cap)), "cap")
        t1 = time.perf_counter()
        p.synchronize()
        t2 = time.perf_counter()
        ms, nl = p.last_propagate_ms()
        print(f"{mode:9s} enqueue {(t1 - t0) * 1e3:6.2f} ms, + wait {(t2 - t1) * 1e3:6.2f} ms = {(t2 - t0) / steps * 1e6:6.2f} us per step; device (run's events) {ms / steps * 1e3:6.2f} us per step, {nl} launches")
