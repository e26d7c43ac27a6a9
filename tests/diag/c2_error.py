import sys; sys.path[:0] = ['.', 'tests/golden']
import numpy as np
import opticomlib_amd as oa
from opticomlib_amd import workloads
from opticomlib_amd.typing import gv, optical_signal
gv(**workloads.BENCH_GV)
g = np.load('tests/golden/c2_full_strided.npz')
y = oa.FIBER(optical_signal(workloads.qpsk_field(1 << 20, seed=2024)), length=125, h=0.125, **workloads.SMF).signal
print("max|d|/peak at the strided samples:", np.max(np.abs(y[:, ::257] - g["samples"])) / np.max(np.abs(g["samples"])))
y2 = np.abs(y.astype(np.complex128))**2
print("power rel diff:", np.mean(y2, axis=-1) / g["power"] - 1, "energy rel diff:", np.sum(y2) / float(g["energy"]) - 1)
