#!/usr/bin/env python3
"""GPU-box diagnostic: run every layer of the HIP path against NumPy / the oracle and print a
table of errors (does not stop at the first failure).  Test infrastructure, not product."""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden")]

from opticomlib_amd import _lib, FIBER, DBP, DM, gv, optical_signal  # noqa: E402
from oracle import ssfm_numpy as orc  # noqa: E402
from cases import CASES, case_dt, case_input  # noqa: E402


def relmax(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def section(name):
    print(f"\n=== {name}", flush=True)


def fft_check(prec, ks):
    cd = np.complex64 if prec == _lib.C64 else np.complex128
    for k in ks:
        n = 1 << k
        try:
            rng = np.random.default_rng(k)
            x = (rng.standard_normal((2, n)) + 1j * rng.standard_normal((2, n)))
            p = _lib.Plan(n, 2, prec)
            p.set_field(x.astype(cd))
            X = p.debug_fft()
            ref = np.fft.fft(x.astype(cd).astype(np.complex128), axis=-1)
            e = np.linalg.norm(X - ref) / np.linalg.norm(ref)
            # identity round trip through apply_transfer(H=1)
            p.set_field(x.astype(cd))
            p.apply_transfer(np.ones(n, dtype=cd))
            y = p.get_field()
            e2 = np.linalg.norm(y - x.astype(cd)) / np.linalg.norm(x)
            print(f"fft  prec={prec} n=2^{k:<2d} relL2(fft)={e:.3e}  relL2(fft->ifft)={e2:.3e}", flush=True)
            p.close()
        except Exception:
            print(f"fft  prec={prec} n=2^{k} FAILED"); traceback.print_exc()


def golden_check():
    for name, case in CASES.items():
        try:
            n = case["inp"][2][-1] if isinstance(case["inp"][2], tuple) else case["inp"][2]
            if n & (n - 1):
                continue
            g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
            gv(**case["gv"])
            sig, noi = case_input(case)
            x = optical_signal(sig) if noi is None else optical_signal(sig, noi)
            kw = dict(case["kw"])
            f = case["func"]
            t = time.time()
            if f in ("FIBER", "DBP"):
                fn = FIBER if f == "FIBER" else DBP
                if kw.get("return_steps"):
                    z, A_z = fn(x, **kw)
                    print(f"{name:28s} z_err={np.max(np.abs(z - g['z'])):.2e} A_z relmax={relmax(A_z, g['A_z']):.3e}")
                    continue
                y = fn(x, **kw)
                msg = f"{name:28s} relmax={relmax(y.signal, g['out']):.3e}"
                if "z" in g:
                    z, _ = fn(x, return_steps=True, **kw)
                    msg += f" steps gpu={len(z) - 1} ref={len(g['z']) - 1}"
                    m = min(len(z), len(g["z"]))
                    msg += f" z_err={np.max(np.abs(z[:m] - g['z'][:m])):.2e}"
                print(msg + f"  [{time.time() - t:.2f}s]", flush=True)
            elif f == "FIBER+DBP":
                mid = FIBER(x, **kw)
                out = DBP(mid, **kw)
                print(f"{name:28s} mid relmax={relmax(mid.signal, g['mid']):.3e} out relmax={relmax(out.signal, g['out']):.3e}")
            elif f == "DM":
                r = DM(x, **kw)
                y = r[0] if kw.get("retH") else r
                msg = f"{name:28s} relmax={relmax(y.signal, g['out']):.3e}"
                if "out_noise" in g:
                    msg += f" noise relmax={relmax(y.noise, g['out_noise']):.3e}"
                print(msg)
            elif f == "TWIN":
                y = FIBER(x, precision="complex128", **kw)
                want = g["A_last"] * np.exp(-(kw["alpha"] / 4.343) * g["z"][-1] / 2)
                print(f"{name:28s} c128 relmax={relmax(y.signal, want):.3e}")
        except Exception:
            print(f"{name} FAILED"); traceback.print_exc()


def big_check():
    gv(sps=16, R=32e9)
    from opticomlib_amd.workloads import qpsk_field
    for k, steps in ((16, 20), (18, 10), (20, 5)):
        try:
            a = qpsk_field(1 << k, seed=2024)
            kw = dict(length=steps * 0.125, h=0.125, alpha=0.2, beta_2=-21.7, beta_3=0.13, gamma=1.3)
            t = time.time(); y = FIBER(optical_signal(a), **kw).signal; tg = time.time() - t
            t = time.time(); r = orc.fiber_c64(a, gv.dt, **kw); tc = time.time() - t
            print(f"big 2x2^{k} {steps} steps: relmax={relmax(y, r):.3e}  gpu_call={tg:.3f}s cpu={tc:.2f}s", flush=True)
        except Exception:
            print(f"big 2^{k} FAILED"); traceback.print_exc()


if __name__ == "__main__":
    print("devices:", _lib.device_count())
    section("FFT c64"); fft_check(_lib.C64, range(8, 21))
    section("FFT c128"); fft_check(_lib.C128, range(8, 21))
    section("golden"); golden_check()
    section("big"); big_check()
