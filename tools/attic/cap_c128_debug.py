"""complex128 capture: which snapshots differ from plain runs (round 5 debugging aid)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import opticomlib_amd as oa
from opticomlib_amd import _lib, workloads
from opticomlib_amd.typing import gv
gv(**workloads.BENCH_GV)
for log2n, npol, lanes, nsteps, every, prec in ((16, 1, 1, 10, 3, _lib.C128), (16, 2, 2, 10, 3, _lib.C128), (12, 1, 1, 30, 7, _lib.C128), (20, 2, 2, 6, 2, _lib.C128), (10, 1, 1, 12, 5, _lib.C64)):
    n = 1 << log2n
    os.environ["SSFM_LANES"] = str(lanes)
    dt = np.complex128 if prec == _lib.C128 else np.complex64
    a = workloads.qpsk_field(n, seed=3, n_pol=2)[:npol].astype(dt)
    D = oa.devices.linear_operator(n, gv.dt, 0.2, -21.7, 0.13)
    hs = np.full(nsteps, 0.05, np.float64 if prec == _lib.C128 else np.float32)
    p = _lib.Plan(n, npol, prec)
    p.set_linear_operator(D)
    for log in (False, True, False):
        p.set_field(a)
        cap = p.propagate_fixed_capture(1.3, hs, every=every, scalars=log)
        end = p.get_field()
        line = [f"2^{log2n} x {npol} lanes {p.lanes} prec {prec} log {log}: steps {list(cap['steps'])} engine {p.last_run_info()['engine']}:"]
        for k, s_ in enumerate(cap["steps"]):
            if s_ == 0:
                want = a
            else:
                p.set_field(a); p.propagate_fixed(1.3, hs[:s_]); p.synchronize(); want = p.get_field()
            line.append("%.1e" % (np.abs(cap["fields"][k] - want).max() / np.abs(want).max()))
        line.append("| end %.1e" % (np.abs(end - cap["fields"][-1]).max()))
        if log:
            pw = np.mean(np.abs(cap["fields"].astype(np.complex128)) ** 2, axis=-1)
            line.append("| power %.1e peak %.1e" % (np.max(np.abs(cap["power"][cap["steps"]] / pw - 1)), np.max(np.abs(cap["peak"][cap["steps"]] / np.max(np.abs(cap["fields"]) ** 2, axis=-1) - 1))))
        print(" ".join(line), flush=True)
    p.close()
