"""Randomised stress of the capture beside the run (ssfm_propagate_fixed_capture): random plan sizes, polarisations, lanes, schedules, strides, with and
without the scalar log, plans reused for several capture runs back to back (the helper thread of one run is joined by the next call).  Every snapshot
against a plain run of that many steps on the same plan (bit for bit where both take the two-kernel engine), the end field, the log against the
snapshots' own power and peak.        python tests/diag/capture_stress.py [cases] [seed]  ->  gpurun_out/r05_capture_stress.txt"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import opticomlib_amd as oa
from opticomlib_amd import _lib, workloads
from opticomlib_amd.typing import gv

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
gv(**workloads.BENCH_GV)
bad, exact, close, worst = [], 0, 0, 0.0
t0 = time.time()
for c in range(cases):
    log2n = int(rng.choice([10, 12, 14, 15, 16, 17, 18, 19, 20], p=[.1, .1, .15, .1, .15, .1, .1, .1, .1]))
    n, npol = 1 << log2n, int(rng.integers(1, 3))
    prec = _lib.C64 if rng.random() < 0.8 else _lib.C128
    nsteps = int(rng.integers(3, 400 if log2n <= 16 else 60))
    every = int(rng.integers(1, max(2, nsteps // 2)))
    os.environ["SSFM_LANES"] = str(int(rng.integers(1, 3)))
    dtype = np.complex64 if prec == _lib.C64 else np.complex128
    a = workloads.qpsk_field(n, seed=int(rng.integers(1 << 30)), n_pol=2)[:npol].astype(dtype)
    D = oa.devices.linear_operator(n, gv.dt, 0.2, -21.7, 0.13)
    hs = np.where(rng.random(nsteps) < 0.9, 0.05, 0.02).astype(np.float32 if prec == _lib.C64 else np.float64)
    p = _lib.Plan(n, npol, prec)
    try:
        p.set_linear_operator(D)
        caps = []
        for rep in range(int(rng.integers(1, 4))):               # back to back on one plan, the log on or off
            p.set_field(a)
            caps.append(p.propagate_fixed_capture(1.3, hs, every=every, scalars=bool(rng.random() < 0.4)))
        end = p.get_field()
        cap = caps[0]
        why = []
        ok = np.array_equal(cap["fields"][0], a)
        if not ok: why.append("input")
        for other in caps[1:]:
            same_kernels = ("power" in other) == ("power" in cap)
            d = np.abs(other["fields"] - cap["fields"]).max() / np.abs(a).max()
            tol_e = (6e-4 if nsteps > 100 else 4e-5) if prec == _lib.C64 else 1e-10
            if not (d == 0 if same_kernels else d < tol_e): ok = False; why.append(f"rerun {d:.1e} same_kernels={same_kernels}")
        idx = sorted(set(int(i) for i in rng.integers(1, len(cap["steps"]), size=min(4, len(cap["steps"]) - 1))))
        for k in idx:
            s_ = int(cap["steps"][k])
            p.set_field(a); p.propagate_fixed(1.3, hs[:s_]); p.synchronize()
            want = p.get_field()
            if p.last_run_info()["engine"] == "two_kernel" and "power" not in cap:
                e = float(np.abs(cap["fields"][k] - want).max()); exact += 1
                if e != 0.0: ok = False; why.append(f"snapshot {k} not bit-identical: {e:.1e}")
            else:
                e = float(np.abs(cap["fields"][k] - want).max() / np.abs(want).max()); close += 1; worst = max(worst, e)
                tol_e = (6e-4 if s_ > 100 else 4e-5) if prec == _lib.C64 else 1e-10       # (two engines, each within the suite's tolerance of the oracle)
                if not e < tol_e: ok = False; why.append(f"snapshot {k} (step {s_}) {e:.1e} engine {p.last_run_info()['engine']} log={'power' in cap}")
        for cp in caps:
            if "power" in cp:
                pw = np.mean(np.abs(cp["fields"].astype(np.complex128)) ** 2, axis=-1)
                if not np.allclose(cp["power"][cp["steps"]], pw, rtol=2e-5): ok = False; why.append("power %.1e" % np.max(np.abs(cp["power"][cp["steps"]] / pw - 1)))
        if not ok:
            bad.append((c, log2n, npol, prec, nsteps, every, os.environ["SSFM_LANES"], why))
    finally:
        p.close()
out = [f"# {cases} random capture configurations (2^10 ... 2^20 samples, 1-2 polarisations, complex64 / complex128, 1-2 lanes, 3 ... 400 steps, stride 1 ... nsteps / 2, 1-3 capture runs back to back per plan, the scalar log on in 40 %), {time.time() - t0:.0f} s",
       f"snapshots compared bit for bit with a plain run of that many steps: {exact}; compared within the engines' tolerance (the plain run on a one-launch engine, or the log's kernels): {close}, worst {worst:.1e}",
       f"failures: {len(bad)} {bad}"]
print("\n".join(out))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "r05_capture_stress.txt"), "w").write("\n".join(out) + "\n")
sys.exit(1 if bad else 0)
