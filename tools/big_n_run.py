"""One fixed-step run of a 2^LOG2N x 2 complex64 field (C2's fibre and step size), for the profiler:  python3 tools/big_n_run.py LOG2N [STEPS]
prints the wall-clock step time of the second run."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from opticomlib_amd import _lib, devices, workloads

k = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = 1 << k
dt = 1.0 / (16 * 32e9)
rng = np.random.default_rng(k)
a = ((rng.standard_normal((2, n)) + 1j * rng.standard_normal((2, n))) * 0.03).astype(np.complex64)
p = _lib.Plan(n, 2, _lib.C64)
p.set_linear_operator(devices.linear_operator(n, dt, 0.2, -21.7, 0.13))
p.set_field(a)
hs = np.full(steps, 0.125, np.float32)
p.propagate_fixed(1.3, hs); p.synchronize()
t = time.perf_counter(); p.propagate_fixed(1.3, hs); p.synchronize(); us = (time.perf_counter() - t) / steps * 1e6
print(f"2^{k} x 2 [{p.last_run_info()['engine']}]: {us:.1f} us per step, step_frac {32 * n / us * 1e6 / 8e12:.3f}", flush=True)
p.close()
