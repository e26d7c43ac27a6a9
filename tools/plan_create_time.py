"""Plan creation time and the lanes a plan ends up with, large plans first in the process (dev aid).  ORDER="c128:22,c64:20,..." """
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from opticomlib_amd import _lib, devices
dt = 1.0 / (16 * 32e9)
for spec in os.environ.get("ORDER", "c128:22,c128:22,c128:21,c64:22,c64:20,c128:20").split(","):
    name, k = spec.split(":"); k = int(k); prec = _lib.C128 if name == "c128" else _lib.C64
    t = time.perf_counter(); p = _lib.Plan(1 << k, 2, prec); p.synchronize(); el = time.perf_counter() - t
    p.set_linear_operator(devices.linear_operator(1 << k, dt, 0.2, -21.7, 0.13, prec))
    rng = np.random.default_rng(k)
    p.set_field(((rng.standard_normal((2, 1 << k)) + 1j * rng.standard_normal((2, 1 << k))) * 0.03).astype(np.complex128 if name == "c128" else np.complex64))
    hs, _ = devices.step_schedule(100 * 0.1, 0.1, prec)
    p.propagate_fixed(1.3, hs); p.synchronize()
    t = time.perf_counter(); p.propagate_fixed(1.3, hs); p.synchronize(); tf = (time.perf_counter() - t) / hs.size
    info = p.last_run_info()
    print(f"{name} 2^{k} x 2: created in {el*1e3:7.1f} ms; {tf*1e6:6.1f} us per step;", {k_: info[k_] for k_ in info if "lane" in k_}, flush=True)
    p.close()
