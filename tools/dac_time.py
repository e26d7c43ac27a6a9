import sys, time; sys.path.insert(0, '.')
import numpy as np
import opticomlib_amd as oa
from opticomlib_amd.typing import gv
from scipy import signal as sg
gv(sps=16, R=10e9)
bits = oa.PRBS(15, len=1 << 16).data
for shape in ("nrz", "gaussian"):
    oa.DAC(bits, pulse_shape=shape)
    t = time.perf_counter(); y = oa.DAC(bits, pulse_shape=shape).signal; el = time.perf_counter() - t
    xu = np.zeros(bits.size * 16); xu[8::16] = bits
    tt = np.linspace(-(bits.size - 4) / 2, (bits.size - 4) / 2, (bits.size - 4) * 16 + 1)
    h = np.where((tt >= -0.5) & (tt < 0.5), 1.0, 0.0) if shape == "nrz" else np.exp(-(2 * np.sqrt(np.log(2)) * (1 + 0j) * tt) ** 2)
    t = time.perf_counter(); r = sg.fftconvolve(xu, h, mode="same"); el2 = time.perf_counter() - t
    print(f"DAC {shape}, 2^16 bits x 16 = 2^20 samples: {el*1e3:.1f} ms (scipy fftconvolve alone on this host: {el2*1e3:.1f} ms), max diff {np.max(np.abs(y - r)):.1e}")
