"""Error against the float64 restatement after 100 steps, per engine of the power-of-two path (which knob set ran it), same field."""
import os, sys, subprocess, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

if len(sys.argv) > 1:
    import opticomlib_amd as oa
    from opticomlib_amd import workloads, _lib
    from opticomlib_amd.devices import get_plan
    from opticomlib_amd.typing import optical_signal, gv
    from oracle import ssfm_numpy as orc
    gv(**workloads.BENCH_GV)
    out = {}
    for n in (4096, 8192, 16384, 65536):
        a = workloads.qpsk_field(1 << 16, seed=5, power_w=4e-3, n_pol=1)[0, :n]
        kw = dict(length=50.0, h=0.5, **workloads.SMF)
        y = oa.FIBER(optical_signal(a), **kw).signal
        eng = get_plan(n, 1, _lib.C64, 0).last_run_info()["engine"]
        r = orc.fiber_c128(a, gv.dt, **kw)
        out[n] = (eng, float(np.max(np.abs(y - r)) / np.max(np.abs(r))))
    print(json.dumps(out))
else:
    for env in ({}, {"SSFM_MEDIUM": "0"}, {"SSFM_SMALL": "0"}, {"SSFM_PHASE_TABLE": "0"}, {"SSFM_MEDIUM": "0", "SSFM_PHASE_TABLE": "0"}, {"SSFM_MEDIUM": "0", "SSFM_SMALL": "0"}):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True)
        print(env, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
