"""The one-launch chirp-z engine of lengths 2048 < n <= 65536 (k_medium_chirp, complex64 callers, fixed steps) against the oracle and the five-launch form:
error after 100 steps and time per step.      python tests/diag/chirp_medium_check.py  ->  profiles/r04_chirp_medium.txt"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import opticomlib_amd as oa
from opticomlib_amd import workloads, _lib
from opticomlib_amd.devices import get_plan
from opticomlib_amd.typing import optical_signal, gv
from oracle import ssfm_numpy as orc


def relmax(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / np.max(np.abs(np.asarray(b))))


def timed(x, reps, **kw):
    oa.FIBER(x, **kw)
    t0 = time.perf_counter()
    for _ in range(reps):
        y = oa.FIBER(x, **kw)
    return (time.perf_counter() - t0) / reps, y


if __name__ == "__main__":
    gv(**workloads.BENCH_GV)
    out = ["# n x rows | engine | err vs oracle complex64 (100 steps) | err vs float64 | us per step one-launch (1000 steps, wall) | us per step on the general complex128 line (SSFM_MEDIUM=0: four launches per step; five -- 21.4 / 24.3 / 29.1 us at 3000 x 2 / 8176 x 2 / 32752 x 2 -- before ssfm_chirp_line_run) || adaptive (20 km, phi_max 0.002): engine, steps, err vs oracle complex64, us per step one-launch | seven-launch form"]
    for n, npol in ((3000, 2), (4095, 1), (8176, 2), (8176, 1), (15060, 2), (16383, 1), (32752, 2), (32752, 1), (40000, 1), (65536 - 3, 1)):
        a = workloads.qpsk_field(1 << 16, seed=n % 997, power_w=4e-3, n_pol=2)[:npol, :n]
        a = a[0] if npol == 1 else a
        x = optical_signal(a)
        kw = dict(length=50.0, h=0.5, **workloads.SMF)
        os.environ["SSFM_MEDIUM"] = "1"
        y = oa.FIBER(x, **kw)
        eng = getattr(y, "engine", "?")
        M = 1 << (2 * n - 2).bit_length()
        info = get_plan(M, npol, _lib.C64, 0).last_run_info()["engine"] if M * npol <= (1 << 17) else "-"
        e64 = relmax(y.signal, orc.fiber_c64(a, gv.dt, **kw))
        e128 = relmax(y.signal, orc.fiber_c128(a, gv.dt, **kw))
        kw2 = dict(length=500.0, h=0.5, **workloads.SMF)
        t1, _ = timed(x, 2, **kw2)
        os.environ["SSFM_MEDIUM"] = "0"
        t5, y5 = timed(x, 1, **kw2)
        os.environ["SSFM_MEDIUM"] = "1"
        # adaptive: steps and time per step, both forms; the error against the oracle's adaptive run
        kwa = dict(length=20.0, phi_max=0.002, **workloads.SMF)
        ta, ya = timed(x, 1, **kwa)
        za, _ = oa.FIBER(x, return_steps=True, **kwa)                     # (the general path: its z list gives the step count)
        ea = relmax(ya.signal, orc.fiber_c64(a, gv.dt, **kwa))
        infa = get_plan(M, npol, _lib.C64, 0).last_run_info()["engine"] if M * npol <= (1 << 17) else "-"
        os.environ["SSFM_MEDIUM"] = "0"
        ta5, _ = timed(x, 1, **kwa)
        os.environ["SSFM_MEDIUM"] = "1"
        nst = len(za) - 1
        line = (f"{n:6d} x {npol} | {info:13s} | {e64:.2e} | {e128:.2e} | {t1 * 1e6 / 1000:7.2f} | {t5 * 1e6 / 1000:7.2f} || {infa} {nst} steps err {ea:.2e} "
                f"{ta * 1e6 / nst:7.2f} | {ta5 * 1e6 / nst:7.2f}")
        print(line, flush=True)
        out.append(line)
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "gpurun_out", "r04_chirp_medium.txt")
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    open(dst, "w").write("\n".join(out) + "\n")
