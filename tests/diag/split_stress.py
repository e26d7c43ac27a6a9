"""Randomised stress of the split plans (csrc/ssfm_split.hpp, round 6): plans of 2^21 / 2^22 samples made to split (SSFM_SPLIT_ABOVE=20: R = 2 / 4) beside direct plans of the same
shape, random sequences of operations on both -- fixed-step runs (one ... seven step sizes: tables or the operator formed in the launch), adaptive runs, every-step snapshots, DM,
device-to-device transfers, runs on the resident result, a caller writing the field through its device address -- and the results compared after every operation.
    python tests/diag/split_stress.py [count] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from opticomlib_amd import _lib, devices, workloads
from opticomlib_amd.accuracy import tol

count = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
dt = 1.0 / (16 * 32e9)
fails, worst = [], 0.0


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


for case in range(count):
    log2n = int(rng.choice([21, 21, 22]))
    n = 1 << log2n
    rows = int(rng.choice([1, 2, 2, 3]))
    c128 = log2n == 21 and rng.integers(0, 3) == 0
    P, cd, rt = (_lib.C128, np.complex128, np.float64) if c128 else (_lib.C64, np.complex64, np.float32)
    a = (workloads.qpsk_field(n, seed=int(rng.integers(0, 1 << 30)), n_pol=2, power_w=float(rng.choice([1e-3, 5e-3])))[:1] * np.ones((rows, 1))).astype(cd)
    a *= (1 + 0.05 * np.arange(rows))[:, None]
    sign = -1.0 if rng.integers(0, 4) == 0 else 1.0
    fib = dict(alpha=sign * 0.2, beta_2=sign * float(rng.uniform(-25, 25)), beta_3=sign * float(rng.choice([0.0, 0.13])), gamma=sign * float(rng.choice([0.0, 1.3, 2.5])))
    D = devices.linear_operator(n, dt, fib["alpha"], fib["beta_2"], fib["beta_3"], P)
    os.environ["SSFM_SPLIT_ABOVE"] = "20"
    ps = _lib.Plan(n, rows, P)
    os.environ.pop("SSFM_SPLIT_ABOVE")
    pd = _lib.Plan(n, rows, P)
    desc, steps_total = [], 0
    try:
        for p in (ps, pd):
            p.set_linear_operator(D); p.set_field(a)
        for op_no in range(int(rng.integers(1, 4))):
            op = rng.choice(["fixed", "fixed", "adaptive", "snap", "dm", "d2d", "poke"])
            if op == "fixed":
                k = int(rng.integers(1, 8))
                hs = rng.choice(rng.uniform(0.05, 0.4, k), int(rng.integers(2, 7))).astype(rt)
                for p in (ps, pd): p.propagate_fixed(fib["gamma"], hs)
                steps_total += hs.size
                desc.append(f"fixed x{hs.size} ({len(set(hs.tolist()))} sizes)")
            elif op == "adaptive" and fib["gamma"] != 0.0:
                out = [p.propagate_adaptive(fib["gamma"], 0.8, 0.004, False) for p in (ps, pd)]
                if out[0][0] != out[1][0]:
                    fails.append((case, "step counts", out[0][0], out[1][0]))
                steps_total += out[0][0]
                desc.append(f"adaptive ({out[0][0]} steps)")
            elif op == "snap":
                hs = np.full(2, 0.2, rt)
                sn = [np.array(p.propagate_fixed(fib["gamma"], hs, snapshots=True)) for p in (ps, pd)]
                steps_total += 2
                e = rel(sn[0], sn[1])
                if not e < (tol(steps_total) if not c128 else 1e-10):
                    fails.append((case, "snapshots", e))
                desc.append("snapshots x2")
            elif op == "dm" and c128:
                for p in (ps, pd): p.apply_dispersion(dt, -100.0e-24)
                for p in (ps, pd): p.set_linear_operator(D)                       # (DM uses the operator's staging buffer)
                desc.append("DM")
            elif op == "d2d":
                for p in (ps, pd):
                    buf = _lib.DeviceArray((rows, n), cd, 0)
                    p.get_field_device(buf.ptr); p.set_field_device(buf.ptr)
                desc.append("d2d round trip")
            elif op == "poke":
                # a caller that writes the field through its device address (what DM() of the Python mirror does)
                half = (a * 0.5).astype(cd)
                for p in (ps, pd):
                    p.copy_into_field(0, half.ctypes.data, half.nbytes, on_device=False)
                steps_total = 0
                desc.append("field written through its address")
            else:
                continue
            ys, yd = ps.get_field(), pd.get_field()
            e = rel(ys, yd)
            worst = max(worst, e / (tol(max(steps_total, 1)) if not c128 else 1e-10))
            if not e < (tol(max(steps_total, 1)) if not c128 else 1e-10):
                fails.append((case, desc[-1], e))
        es, ed = ps.last_run_info()["engine"], pd.last_run_info()["engine"]
        print(f"{case:3d} 2^{log2n} x {rows} {'c128' if c128 else 'c64 '} gamma {fib['gamma']:+.1f}: {', '.join(desc)} | split [{es}] vs direct [{ed}] {rel(ps.get_field(), pd.get_field()):.2e}", flush=True)
    finally:
        ps.close(); pd.close()
print(f"{count} cases (seed {seed}): failures {len(fails)} {fails[:5]}; worst distance / bound {worst:.3f}")
sys.exit(1 if fails else 0)
