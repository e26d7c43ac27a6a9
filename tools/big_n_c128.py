"""complex128 plans from 2^19 to 2^22 points: us per step and the bytes per second they stand for (tuning aid)."""
import sys, time; sys.path.insert(0, '.')
import numpy as np
from opticomlib_amd import _lib, devices
dt = 1.0 / (16 * 32e9)
for k in (19, 20, 21, 22):
    n = 1 << k
    for pol in (1, 2):
        rng = np.random.default_rng(k)
        a = ((rng.standard_normal((pol, n)) + 1j * rng.standard_normal((pol, n))) * 0.03)
        p = _lib.Plan(n, pol, _lib.C128); p.set_linear_operator(devices.linear_operator(n, dt, 0.2, -21.7, 0.13, _lib.C128)); p.set_field(a)
        hs, _ = devices.step_schedule(200 * 0.1, 0.1, _lib.C128)
        p.propagate_fixed(1.3, hs); p.synchronize()
        t = time.perf_counter(); p.propagate_fixed(1.3, hs); p.synchronize(); tf = (time.perf_counter() - t) / hs.size
        print(f"c128 2^{k} x {pol}: {tf*1e6:6.1f} us/step  -> {n * pol * 32 / tf / 1e12:.2f} TB/s algorithmic (32 B per sample-pol-step)", flush=True)
        p.close()
