"""Per-stage wall times of the example chain PRBS -> DAC -> MZM(LASER) -> FIBER -> PD -> read-back, call after call
(found: read-backs into pageable memory stalled the next transfer by 20 ms; results now land in pooled page-locked buffers)."""
import sys, time; sys.path.insert(0, '.')
import numpy as np
from opticomlib_amd import DAC, FIBER, LASER, MZM, PD, PRBS, gv
bits = 1 << 12; Vpi = 5.0
gv(sps=64, R=10e9, N=bits)
def link():
    T = [time.perf_counter()]
    def lap(): T.append(time.perf_counter())
    tx = PRBS(order=15, len=bits); lap()
    drive = DAC(tx, Vpp=Vpi, offset=-Vpi / 2, pulse_shape="gaussian"); lap()
    cw = LASER(P0=5); lap()
    field = MZM(cw, drive, bias=-Vpi / 2, Vpi=Vpi, loss_dB=3, ER_dB=26); lap()
    out = FIBER(field, length=50, alpha=0.2, beta_2=-20, gamma=2); lap()
    rx = PD(out, BW=0.75 * gv.R, r=1.0, include_noise="all"); lap()
    v = rx.signal + rx.noise; lap()
    return [round((b - a) * 1e3, 2) for a, b in zip(T, T[1:])]
for _ in range(4):
    print("PRBS, DAC, LASER, MZM, FIBER, PD, read [ms]:", link())
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); link(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
