"""Where does the four-step (N1 x N2) engines' distance from the float64 solution come from?  Fixed-step runs of 83 steps at 2^12 ... 2^16 on the two-kernel engine
(SSFM_MEDIUM=0 SSFM_SMALL=0), the one-XCD engine and -- up to 8192 samples -- the one-workgroup engine, each against the float64 solution of the same schedule and
against the oracle.  With SSFM_LIB pointing at a -DSSFM_DIAG_EXACT_TWN=1 build the inter-pass twiddles are rounded once from double instead of formed as a float32 product.
    python tests/diag/fourstep_error.py"""
import os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_cases as fc
from opticomlib_amd import _lib, devices, workloads
from oracle import ssfm_numpy as orc

dt = 1.0 / (16 * 32e9)
kw = dict(alpha=0.186, beta_2=-20.6, beta_3=-0.298, gamma=3.37)
steps = 83
hs = np.full(steps, 0.094, np.float32)
print(f"# lib: {os.environ.get('SSFM_LIB', 'product')}")
for log2n in (12, 13, 14, 16):
    n = 1 << log2n
    a = workloads.qpsk_field(n, seed=7, n_pol=1, power_w=3e-3)[0]
    t = fc.truth_f64(a, dt, hs, dict(length=float(hs.sum()), h=0.094, **kw))
    ref = a.astype(np.complex64)
    D = orc.linear_operator_c64(n, dt, kw["alpha"], kw["beta_2"], kw["beta_3"])
    for h_ in hs:
        ref = orc.ssfm_step_c64(ref, D, np.float32(kw["gamma"]), h_)
    pk = float(np.max(np.abs(t)))
    line = f"2^{log2n} x 1, {steps} steps: oracle-f64 {np.max(np.abs(ref - t)) / pk:.2e} |"
    for name, env in (("two-kernel", {"SSFM_MEDIUM": "0", "SSFM_SMALL": "0"}), ("two-kernel E=16", {"SSFM_MEDIUM": "0", "SSFM_SMALL": "0", "SSFM_E": "16", "SSFM_EF": "16"}),
                      ("one-XCD", {"SSFM_SMALL": "0"}), ("one-workgroup", {"SSFM_MEDIUM": "0"})):
        for k in ("SSFM_MEDIUM", "SSFM_SMALL", "SSFM_E", "SSFM_EF"):
            os.environ.pop(k, None)
        os.environ.update(env)
        p = _lib.Plan(n, 1, _lib.C64)
        try:
            p.set_linear_operator(devices.linear_operator(n, dt, kw["alpha"], kw["beta_2"], kw["beta_3"]))
            p.set_field(a.astype(np.complex64).reshape(1, n))
            p.propagate_fixed(kw["gamma"], hs)
            y = p.get_field()[0]
            eng = p.last_run_info()["engine"]
        finally:
            p.close()
        line += f" {name} [{eng}] HIP-f64 {np.max(np.abs(y - t)) / pk:.2e} HIP-oracle {np.max(np.abs(y - ref)) / pk:.2e} |"
    print(line, flush=True)
