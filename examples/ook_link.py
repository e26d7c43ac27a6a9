"""A 10 Gb/s on-off-keyed link on the MI355X path, end to end:

    PRBS -> DAC (Gaussian pulses) -> MZM(LASER) -> FIBER (50 km SMF, adaptive split step) -> PD -> decisions

the chain of the reference's own example (opticomlib: examples/ook_transmission_fiber_simulation.py) with every
device taken from opticomlib_amd.  Everything between the modulator and the detector output stays in GPU memory.

    python examples/ook_link.py [bits] [length_km]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticomlib_amd import DAC, FIBER, LASER, MZM, PD, PRBS, gv  # noqa: E402

bits = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 12
length = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
Vpi = 5.0
gv(sps=64, R=10e9, N=bits)

def link():
    t0 = time.perf_counter()
    tx = PRBS(order=15, len=bits)
    drive = DAC(tx, Vpp=Vpi, offset=-Vpi / 2, pulse_shape="gaussian")
    field = MZM(LASER(P0=5), drive, bias=-Vpi / 2, Vpi=Vpi, loss_dB=3, ER_dB=26)
    t1 = time.perf_counter()
    out = FIBER(field, length=length, alpha=0.2, beta_2=-20, gamma=2)          # adaptive step, phi_max = 0.01
    rx = PD(out, BW=0.75 * gv.R, r=1.0, include_noise="all")
    v = rx.signal + rx.noise                                                    # the only download
    return tx, out, v, t1 - t0, time.perf_counter() - t1


link()                                  # first call: plans, tables and code objects are created
tx, out, v, t_tx, t_rx = link()

# decisions at the bit centres against the mid-level threshold
samples = v[gv.sps // 2::gv.sps]
threshold = 0.5 * (samples[tx.data == 1].mean() + samples[tx.data == 0].mean())
errors = int(np.count_nonzero((samples > threshold) != (tx.data == 1)))
power_dbm = 10 * np.log10(np.mean(np.abs(out.signal) ** 2) / 1e-3)
print(f"{bits} bits, {bits * gv.sps} samples, {length:g} km: transmitter {1e3 * t_tx:.1f} ms, fibre + detector {1e3 * t_rx:.1f} ms (second call)")
print(f"received power {power_dbm:.2f} dBm, threshold {threshold * 1e3:.2f} mV, {errors} errors in {bits} bits")
