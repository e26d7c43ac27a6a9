"""Where the time of the transmitter of a 2^20-sample link goes (PRBS -> DAC -> MZM(LASER)), call by call."""
import sys, time; sys.path.insert(0, '.')
import numpy as np
import opticomlib_amd as oa
from opticomlib_amd import _lib
from opticomlib_amd.typing import gv
gv(sps=16, R=10e9, N=1 << 16)

def timed(label, f, reps=5):
    f()
    _lib.synchronize_all() if hasattr(_lib, "synchronize_all") else None
    t = time.perf_counter()
    for _ in range(reps):
        r = f()
    el = (time.perf_counter() - t) / reps
    print(f"{label:34s} {el * 1e3:8.2f} ms")
    return r

bits = timed("PRBS(15, 2^16 bits)", lambda: oa.PRBS(15, len=1 << 16))
drive = timed("DAC gaussian", lambda: oa.DAC(bits, Vpp=5.0, offset=-2.5, pulse_shape="gaussian"))
timed("DAC nrz", lambda: oa.DAC(bits, Vpp=5.0, offset=-2.5))
timed("DAC rcos", lambda: oa.DAC(bits, Vpp=5.0, offset=-2.5, pulse_shape="rcos", beta=0.3))
cw = timed("LASER(P0)", lambda: oa.LASER(P0=3))
timed("LASER(P0, df)", lambda: oa.LASER(P0=3, df=1e9))
timed("LASER(P0, lw)", lambda: oa.LASER(P0=3, lw=1e5))
timed("LASER(P0, lw, rin, rng=device)", lambda: oa.LASER(P0=3, lw=1e5, rin=-150, rng="device"))
mod = timed("MZM(cw, drive)", lambda: oa.MZM(cw, drive, bias=-2.5, Vpi=5.0))
timed("whole transmitter", lambda: oa.MZM(oa.LASER(P0=3), oa.DAC(oa.PRBS(15, len=1 << 16), Vpp=5.0, offset=-2.5, pulse_shape="gaussian"), bias=-2.5, Vpi=5.0))
out = timed("FIBER 80 km, h = 1 km", lambda: oa.FIBER(mod, length=80, alpha=0.2, beta_2=-21.7, gamma=1.3, h=1.0))
t = time.perf_counter(); y = out.signal; print(f"{'read .signal':34s} {(time.perf_counter() - t) * 1e3:8.2f} ms")
