"""Per-kernel time vs rows per launch, single stream (dev aid)."""
import os, sys
import numpy as np
os.environ["SSFM_LANES"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from opticomlib_amd import _lib, devices, workloads
n = 1 << 20; dt = 1.0 / (16 * 32e9)
hs, _ = devices.step_schedule(25.0, 0.125)
D = devices.linear_operator(n, dt, 0.2, -21.7, 0.13)
for rows in (1, 2, 4, 8, 16):
    a = np.concatenate([workloads.qpsk_field(n, seed=s) for s in range((rows + 1) // 2)])[:rows].astype(np.complex64)
    p = _lib.Plan(n, rows, _lib.C64); p.set_linear_operator(D); p.set_field(a)
    p.propagate_fixed(1.3, hs); p.synchronize()
    p.set_profiling(2); p.propagate_fixed(1.3, hs); p.synchronize(); kt2 = p.kernel_times()
    p.set_profiling(1); p.propagate_fixed(1.3, hs); p.synchronize(); kt1 = p.kernel_times()
    pooled = sum(v[1] for v in kt2.values()) / sum(v[0] for v in kt2.values()) * 1e3
    d = {k: v[1] / v[0] * 1e3 for k, v in kt1.items()}
    print(f"rows={rows:2d} (WGs per launch {rows*256}): pooled {pooled:.2f} us/launch; dense events k_time {d['k_time']:.2f} k_freq {d['k_freq']:.2f}; per row {pooled/rows:.2f}", flush=True)
    p.close()
