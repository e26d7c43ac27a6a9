import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticomlib_amd import _lib
for k in (12, 16, 20):
    n = 1 << k
    rng = np.random.default_rng(k)
    x = (rng.standard_normal((2, n)) + 1j * rng.standard_normal((2, n))).astype(np.complex64)
    p = _lib.Plan(n, 2, _lib.C64); p.set_field(x)
    X = p.debug_fft(); ref = np.fft.fft(x.astype(np.complex128), axis=-1)
    e1 = np.linalg.norm(X - ref) / np.linalg.norm(ref)
    p.set_field(x); H = np.ones(n, np.complex64)
    for _ in range(200): p.apply_transfer(H)
    y = p.get_field()
    e2 = np.linalg.norm(y - x) / np.linalg.norm(x)
    en = np.sum(np.abs(y.astype(np.complex128))**2) / np.sum(np.abs(x.astype(np.complex128))**2) - 1
    print(f"n=2^{k}: fft relL2 {e1:.3e}; after 200 fft->ifft round trips relL2 {e2:.3e}, energy drift {en:.2e}")
    p.close()
