"""The K = 1 case of lane_speed_probe.py under AMD_LOG_LEVEL=4: which hardware queue the dispatches of the slow plan's streams go to.
    AMD_LOG_LEVEL=4 python tools/attic/lane_speed_hwq.py 2> log; python tools/attic/lane_speed_hwq.py parse log"""
import os, sys, re, collections
if len(sys.argv) > 2 and sys.argv[1] == "parse":
    phase, per = "start", collections.OrderedDict()
    for line in open(sys.argv[2], errors="replace"):
        m = re.search(r"PROBE-MARK (\S+)", line)
        if m:
            phase = m.group(1)
        m = re.search(r"HWq=(0x[0-9a-f]+)", line)
        if m:
            sw = re.search(r"SWq=(0x[0-9a-f]+)", line)
            key = (phase, sw.group(1) if sw else "?", m.group(1))
            per[key] = per.get(key, 0) + 1
    for k, v in per.items():
        print(k, v)
    sys.exit(0)
import time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import opticomlib_amd as oa
from opticomlib_amd import _lib, gv, workloads
gv(**workloads.BENCH_GV)
n = 1 << 19
a = workloads.qpsk_field(n, seed=4).astype(np.complex64)
D = oa.devices.linear_operator(n, gv.dt, 0.2, -21.7, 0.13)
hs = np.full(40, 0.125, np.float32)
D14 = oa.devices.linear_operator(1 << 14, gv.dt, 0.2, -21.7, 0.13)
keep = []
def mark(s):
    sys.stderr.write(f"\nPROBE-MARK {s}\n"); sys.stderr.flush()
for K in range(3):
    os.environ["SSFM_LANES"] = "2"
    mark(f"p{K}-create")
    p = _lib.Plan(n, 2, _lib.C64)
    p.set_linear_operator(D); p.set_field(a)
    p.propagate_fixed(1.3, hs); p.synchronize()
    mark(f"p{K}-run")
    t0 = time.perf_counter(); p.propagate_fixed(1.3, hs); p.synchronize(); t = time.perf_counter() - t0
    mark(f"p{K}-after")
    print(f"K={K}: {t / 40 * 1e6:6.2f} us per step", flush=True)
    p.close()
    os.environ["SSFM_LANES"] = "1"
    q = _lib.Plan(1 << 14, 1, _lib.C64)
    q.set_linear_operator(D14); q.set_field(workloads.qpsk_field(1 << 14, seed=K, n_pol=1)); q.propagate_fixed(1.3, hs[:3]); q.synchronize()
    keep.append(q)
