"""Round 4: what the order dependence of the (since removed) two-stream fused adaptive engine was.
    python tools/attic/order_dependence_probe.py [giveup] [dummies=K]
`giveup`: first do what test_one_launch_adaptive_run_of_any_length_gives_up_cleanly did (a 512-point x 2 complex128 plan made with
SSFM_FUSED_PATIENCE_TICKS=-1, run once step by step and once through the one-launch chirp-z kernel that gives up).  dummies=K: K extra
high-priority streams' worth of plans alive.  Then an adaptive run of 2^19 x 2 with SSFM_ADAPT_FUSED_LANES=1 and its launch count:
4 per step = the two lanes ran, 3 per step = the first hand-over between the lanes ran out of its 20 ms and the plan fell back.
With AMD_LOG_LEVEL=4 the runtime logs the hardware queue of every dispatch (HWq=...): tools/attic/hwq_of_lanes.py reads them."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import opticomlib_amd as oa
from opticomlib_amd import _lib, gv, optical_signal, workloads

args = sys.argv[1:]
gv(**workloads.BENCH_GV)
keep = []
for a in args:
    if a.startswith("dummies="):
        for k in range(int(a.split("=")[1])):
            q = _lib.Plan(1 << 14, 1, _lib.C64)                  # one high-priority stream each, USED (the runtime maps a stream to a hardware queue at its first launch)
            q.set_linear_operator(oa.devices.linear_operator(1 << 14, gv.dt, 0.2, -21.7, 0.13))
            q.set_field(workloads.qpsk_field(1 << 14, seed=k, n_pol=1))
            q.propagate_fixed(1.3, np.full(3, 0.125, np.float32)); q.synchronize()
            keep.append(q)
if "giveup" in args:
    x = optical_signal(workloads.qpsk_field(1 << 11, seed=5, power_w=8e-3)[:, :200])
    kw = dict(length=8.0, phi_max=0.004, **workloads.SMF)
    os.environ["SSFM_FUSED_PATIENCE_TICKS"] = "-1"
    os.environ["SSFM_CHIRP_SMALL"] = "0"
    ref = oa.FIBER(x, **kw).signal
    os.environ["SSFM_CHIRP_SMALL"] = "1"
    y = oa.FIBER(x, **kw).signal
    del os.environ["SSFM_FUSED_PATIENCE_TICKS"], os.environ["SSFM_CHIRP_SMALL"]
    print("give-up sequence done, identical:", bool(np.array_equal(y, ref)), flush=True)
n = 1 << 19
a = workloads.qpsk_field(n, seed=109, power_w=10e-3)
D = oa.devices.linear_operator(n, gv.dt, 0.2, -21.7, 0.13)
os.environ["SSFM_LANES"] = "2"; os.environ["SSFM_ADAPT_FUSED"] = "1"; os.environ["SSFM_ADAPT_FUSED_LANES"] = "1"
sys.stderr.write("PROBE-MARK two-lane plan begins\n"); sys.stderr.flush()
p = _lib.Plan(n, 2, _lib.C64)
p.set_linear_operator(D)
for rep in range(2):
    p.set_field(a)
    steps, z, _ = p.propagate_adaptive(1.3, 1.0, 0.004, False)
    print(f"rep {rep}: {steps} steps, {p.last_propagate_ms()[1]} launches ({p.last_propagate_ms()[1] / steps:.2f} per step)", flush=True)
p.close()
