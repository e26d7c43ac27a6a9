"""the one-XCD adaptive engine (k_medium_adapt) sits 1.5 ... 1.8 x as far from the float64 solution as the oracle does at 67-83 steps
(profiles/r05_final_fuzz.txt).  Which term carries it?  The same runs through every adaptive engine of the library and through a fixed-step replay of the run's
OWN schedule (tabulated exp(D~ h) instead of the in-kernel one), each against the float64 solution of ITS OWN schedule and of the oracle's.
    python tests/diag/medium_adapt_terms.py [count] [seed]"""
import os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_cases as fc
import opticomlib_amd as oa
from opticomlib_amd import _lib, workloads
from opticomlib_amd.typing import gv
from oracle import ssfm_numpy as orc

count = int(sys.argv[1]) if len(sys.argv) > 1 else 400
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
gv(**workloads.BENCH_GV)
VARIANTS = [("medium_adaptive (one launch, E=8)", {}),
            ("two launches per step, fused (E=8)", {"SSFM_MEDIUM_ADAPT": "0"}),
            ("three launches per step (E=8)", {"SSFM_MEDIUM_ADAPT": "0", "SSFM_ADAPT_FUSED": "0"}),
            ("three launches per step, E=16", {"SSFM_MEDIUM_ADAPT": "0", "SSFM_ADAPT_FUSED": "0", "SSFM_E": "16", "SSFM_EF": "16"}),
            ("two launches per step, E=16", {"SSFM_MEDIUM_ADAPT": "0", "SSFM_E": "16", "SSFM_EF": "16"})]
KNOBS = ("SSFM_MEDIUM_ADAPT", "SSFM_ADAPT_FUSED", "SSFM_E", "SSFM_EF", "SSFM_MEDIUM", "SSFM_SMALL")


def rel(a, b, pk):
    return float(np.max(np.abs(a - b))) / pk


for i, n, npol, kw, a, pow2 in fc.cases(count, seed):
    if not pow2 or "phi_max" not in kw or not (4096 <= n <= 16384) or kw.get("gamma", 0.0) == 0.0:
        continue
    a2 = np.atleast_2d(a).astype(np.complex64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        zr, Ar = orc.fiber_c64(a, gv.dt, return_steps=True, **kw)
    hs_o = np.diff(zr.astype(np.float32)).astype(np.float32)
    if not (40 <= len(hs_o) <= 120):
        continue
    t_o = np.atleast_2d(fc.truth_f64(a, gv.dt, hs_o, kw))
    pk = float(np.max(np.abs(t_o)))
    e_ot = rel(np.atleast_2d(Ar[-1]), t_o, pk)
    print(f"== case {i}: {n} x {npol}, {len(hs_o)} steps, oracle-float64 {e_ot:.2e}  {kw}", flush=True)
    D = oa.devices.linear_operator(n, gv.dt, kw.get("alpha", 0.0), kw.get("beta_2", 0.0), kw.get("beta_3", 0.0))
    for name, env in VARIANTS:
        for k in KNOBS:
            os.environ.pop(k, None)
        os.environ.update(env)
        p = _lib.Plan(n, a2.shape[0], _lib.C64)
        try:
            p.set_linear_operator(D)
            p.set_field(a2)
            steps, z, _ = p.propagate_adaptive(float(np.float32(kw["gamma"])), float(np.float32(kw["length"])), float(np.float32(kw["phi_max"])), False)
            y = p.get_field()
            eng = p.last_run_info()["engine"]
            hs = np.diff(np.asarray(z, dtype=np.float32)).astype(np.float32)
            t_own = np.atleast_2d(fc.truth_f64(a, gv.dt, hs, kw))
            # the run's own schedule replayed with fixed steps: the tabulated operator instead of the in-kernel exp(D~ h)
            p.set_field(a2)
            p.propagate_fixed(float(np.float32(kw["gamma"])), hs)
            yf = p.get_field()
            engf = p.last_run_info()["engine"]
        finally:
            p.close()
        print(f"   {name:42s} [{eng:20s}] {steps:3d} steps: HIP-oracle {rel(y, np.atleast_2d(Ar[-1]), pk):.2e}  HIP-f64(oracle's schedule) {rel(y, t_o, pk):.2e}  "
              f"HIP-f64(own schedule) {rel(y, t_own, pk):.2e} | fixed-step replay [{engf}] HIP-f64(own) {rel(yf, t_own, pk):.2e}  max|h-h_oracle|/h {np.max(np.abs(hs[:min(len(hs), len(hs_o))] - hs_o[:min(len(hs), len(hs_o))]) / hs_o[:min(len(hs), len(hs_o))]):.1e}", flush=True)
for k in KNOBS:
    os.environ.pop(k, None)
