import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from opticomlib_amd import _lib
for k in (8, 9, 10, 12, 14, 16, 20):
    n = 1 << k
    rng = np.random.default_rng(k)
    x = (rng.standard_normal((2, n)) + 1j * rng.standard_normal((2, n))).astype(np.complex64)
    p = _lib.Plan(n, 2, _lib.C64)
    p.set_field(x)
    H = np.ones(n, dtype=np.complex64)
    e0 = np.sum(np.abs(x.astype(np.complex128)) ** 2)
    reps = 300
    for i in range(reps):
        p.apply_transfer(H)
    y = p.get_field()
    e1 = np.sum(np.abs(y.astype(np.complex128)) ** 2)
    print(f"n=2^{k}: energy drift per fft->ifft round trip = {(e1 / e0 - 1) / reps:.3e}   relL2 err after {reps}: {np.linalg.norm(y - x) / np.linalg.norm(x):.3e}")
    p.close()
