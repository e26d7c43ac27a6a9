"""Split plans (rows of more than 2^22 samples, csrc/ssfm_split.hpp): correctness against the oracle and the direct engine, and step times.
    python tools/split_check.py [quick|full|time]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from opticomlib_amd import _lib, devices, workloads
from oracle import ssfm_numpy as orc

mode = sys.argv[1] if len(sys.argv) > 1 else "quick"
dt = 1.0 / (16 * 32e9)
SMF = workloads.SMF


def rel(a, b):
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


def run(n, pol, hs, prec=_lib.C64, adaptive=None, seed=3, env=None):
    for k in ("SSFM_SPLIT_ABOVE", "SSFM_SPLIT_LOG2M"):
        os.environ.pop(k, None)
    os.environ.update(env or {})
    a = workloads.qpsk_field(n, seed=seed, n_pol=max(pol, 1), power_w=5e-3)[:pol]
    p = _lib.Plan(n, pol, prec)
    try:
        p.set_linear_operator(devices.linear_operator(n, dt, SMF["alpha"], SMF["beta_2"], SMF["beta_3"], prec))
        p.set_field(a.astype(np.complex64 if prec == _lib.C64 else np.complex128))
        if adaptive:
            steps, z, _ = p.propagate_adaptive(SMF["gamma"], adaptive[0], adaptive[1], False)
        else:
            p.propagate_fixed(SMF["gamma"], hs)
            steps, z = len(hs), None
        y = p.get_field()
        info = p.last_run_info()
    finally:
        p.close()
    return a, y, info["engine"], steps, z


if mode in ("quick", "full"):
    for log2n, pol in ((21, 1), (22, 2)):
        n = 1 << log2n
        hs = np.full(3, 0.25, np.float32)
        a, y_split, eng, _, _ = run(n, pol, hs, env={"SSFM_SPLIT_ABOVE": "20"})
        _, y_dir, eng_d, _, _ = run(n, pol, hs)
        line = f"2^{log2n} x {pol} fixed 3 steps: split [{eng}] vs direct [{eng_d}] {rel(y_split, y_dir):.2e}"
        if log2n == 21:
            ref = orc.fiber_c64(a[0] if pol == 1 else a, dt, length=0.75, h=0.25, **SMF)
            line += f"; split vs oracle {rel(y_split, np.atleast_2d(ref)):.2e}, direct vs oracle {rel(y_dir, np.atleast_2d(ref)):.2e}"
        print(line, flush=True)
        a, y_split, eng, s1, z1 = run(n, pol, None, adaptive=(1.0, 0.004), env={"SSFM_SPLIT_ABOVE": "20"})
        _, y_dir, eng_d, s0, z0 = run(n, pol, None, adaptive=(1.0, 0.004))
        print(f"2^{log2n} x {pol} adaptive: split [{eng}] {s1} steps vs direct [{eng_d}] {s0} steps {rel(y_split, y_dir):.2e}; z {np.max(np.abs(np.asarray(z1)[:min(s1, s0)] - np.asarray(z0)[:min(s1, s0)])):.1e}", flush=True)
    # complex128
    hs = np.full(2, 0.25, np.float64)
    a, y_split, eng, _, _ = run(1 << 21, 1, hs, prec=_lib.C128, env={"SSFM_SPLIT_ABOVE": "20"})
    ref = orc.fiber_c128(a[0], dt, length=0.5, h=0.25, **SMF)
    print(f"2^21 x 1 complex128 2 steps: split [{eng}] vs float64 restatement {rel(y_split, np.atleast_2d(ref)):.2e}", flush=True)
    # the real thing
    hs = np.full(2, 0.25, np.float32)
    t = time.time()
    a, y, eng, _, _ = run(1 << 23, 1, hs)
    ref = orc.fiber_c64(a[0], dt, length=0.5, h=0.25, **SMF)
    print(f"2^23 x 1 fixed 2 steps [{eng}] vs oracle {rel(y, np.atleast_2d(ref)):.2e}  ({time.time() - t:.0f} s)", flush=True)
if mode == "full":
    hs = np.full(2, 0.25, np.float32)
    t = time.time()
    a, y, eng, _, _ = run(1 << 24, 2, hs)
    ref = orc.fiber_c64(a, dt, length=0.5, h=0.25, **SMF)
    print(f"2^24 x 2 fixed 2 steps [{eng}] vs oracle {rel(y, ref):.2e}  ({time.time() - t:.0f} s)", flush=True)
    a, y, eng, s1, z1 = run(1 << 23, 1, None, adaptive=(0.6, 0.004))
    zr, Ar = orc.fiber_c64(a[0], dt, length=0.6, phi_max=0.004, return_steps=True, **SMF)
    print(f"2^23 x 1 adaptive [{eng}] {s1} steps (oracle {len(zr) - 1}) vs oracle {rel(y, np.atleast_2d(Ar[-1])):.2e}", flush=True)
if mode in ("time", "full"):
    for log2n in (21, 22, 23, 24):
        for lm in (("20",) if log2n <= 22 else ("20", "21", "22")):
            n = 1 << log2n
            env = {"SSFM_SPLIT_LOG2M": lm} if log2n > 22 else {}
            for k in ("SSFM_SPLIT_ABOVE", "SSFM_SPLIT_LOG2M"):
                os.environ.pop(k, None)
            os.environ.update(env)
            a = workloads.qpsk_field(n, seed=1, n_pol=2).astype(np.complex64)
            p = _lib.Plan(n, 2, _lib.C64)
            p.set_linear_operator(devices.linear_operator(n, dt, SMF["alpha"], SMF["beta_2"], SMF["beta_3"]))
            p.set_field(a)
            hs = np.full(100, 0.125, np.float32)
            p.propagate_fixed(1.3, hs); p.synchronize()
            ts = []
            for _ in range(3):
                t = time.perf_counter(); p.propagate_fixed(1.3, hs); p.synchronize(); ts.append((time.perf_counter() - t) / hs.size * 1e6)
            eng = p.last_run_info()["engine"]
            ta = None
            if log2n >= 22:
                t = time.perf_counter(); steps, _, _ = p.propagate_adaptive(1.3, 3.0, 0.002, False); ta = (time.perf_counter() - t) / max(steps, 1) * 1e6
            p.close()
            us = min(ts)
            print(f"2^{log2n} x 2 [{eng}{', M = 2^' + lm if log2n > 22 else ''}]: {us:8.1f} us per step = {n / us * 1e-3:6.1f} G sample-steps/s, step_frac {32 * n / us * 1e6 / 8e12:.3f}"
                  + (f"; adaptive {ta:8.1f} us per step ({steps} steps)" if ta else ""), flush=True)
