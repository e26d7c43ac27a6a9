import sys; sys.path.insert(0, '.')
import numpy as np
from scipy import signal as sg
from opticomlib_amd import _lib
rng = np.random.default_rng(2)
n = 200000
x = rng.standard_normal((2, n)).cumsum(axis=-1) * 0.01 + rng.standard_normal((2, n))
for fc in (1e-2, 5e-3, 3e-3, 2e-3, 1e-3, 5e-4, 2e-4, 1e-4):
    row = []
    for order in (2, 4, 6, 8):
        sos = sg.bessel(order, 2 * fc, "low", norm="mag", output="sos"); zi = sg.sosfilt_zi(sos)
        want = sg.sosfiltfilt(sos, x, axis=-1); got = _lib.sosfiltfilt(sos, zi, x)
        row.append(float(np.max(np.abs(got - want)) / np.max(np.abs(want))))
    print(f"cutoff/fs {fc:g}: orders 2/4/6/8 -> " + " ".join(f"{e:.1e}" for e in row))
