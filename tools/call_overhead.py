"""Host-side cost of one FIBER() call against its device time, small fields (dev aid)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import opticomlib_amd as oa
from opticomlib_amd import workloads
from opticomlib_amd.typing import gv, optical_signal
gv(sps=64, R=10e9)
for k, npol in ((12, 1), (16, 1), (16, 2)):
    a = workloads.qpsk_field(1 << k, seed=1, n_pol=2, sps=64)[:npol]
    x = optical_signal(a[0] if npol == 1 else a)
    for kw, name in ((dict(length=50, h=5.0), "10 fixed steps"), (dict(length=50), "adaptive (reference default)")):
        kw.update(alpha=0.2, beta_2=-20, gamma=2)
        oa.FIBER(x, **kw)
        t = time.perf_counter()
        for _ in range(20):
            y = oa.FIBER(x, **kw)
        el = (time.perf_counter() - t) / 20
        t = time.perf_counter()
        for _ in range(20):
            y = oa.FIBER(x, **kw).signal
        el2 = (time.perf_counter() - t) / 20
        print(f"2^{k} x {npol}, {name}: {el*1e3:.2f} ms per call, {el2*1e3:.2f} ms with the result read")
