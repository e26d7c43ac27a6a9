import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from opticomlib_amd import FIBER, gv, optical_signal, workloads
gv(**workloads.BENCH_GV)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
a = workloads.qpsk_field(1 << k, seed=1)
e_in = np.sum(np.abs(a) ** 2)
def drift(**kw):
    y = FIBER(optical_signal(a), length=125, h=0.125, **kw).signal
    return np.sum(np.abs(y.astype(np.complex128)) ** 2) / e_in - 1
print("n=2^%d" % k)
print("A  all zero          ", drift())
print("B  beta2             ", drift(beta_2=-21.7))
print("B3 beta2+beta3       ", drift(beta_2=-21.7, beta_3=0.13))
print("C  gamma only(1 step)", drift(gamma=1.3))
print("D  beta2+gamma       ", drift(beta_2=-21.7, gamma=1.3))
print("D128 beta2+gamma c128", (lambda y: np.sum(np.abs(y)**2)/e_in-1)(FIBER(optical_signal(a), length=125, h=0.125, beta_2=-21.7, gamma=1.3, precision="complex128").signal))
