"""Rounding noise of the library's complex64 transforms beside numpy's (pocketfft) on the same data, against float64 (GPU).
    python tests/diag/fft_noise.py  ->  gpurun_out/r05_fft_noise.txt"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from opticomlib_amd import _lib, workloads
import opticomlib_amd as oa
from opticomlib_amd.typing import gv, optical_signal
from oracle import ssfm_numpy as orc
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import chirp_c64_sources as S
from opticomlib_amd.devices import step_schedule

out = []
def say(s):
    print(s, flush=True); out.append(s)

say("# one forward transform and one fft->ifft round trip: relative L2 error against float64, HIP (plan of n points, complex64) | numpy.fft complex64")
for k in (10, 12, 13, 14, 15, 16, 17, 18, 20):
    n = 1 << k
    rng = np.random.default_rng(k)
    x = (rng.standard_normal((1, n)) + 1j * rng.standard_normal((1, n))).astype(np.complex64)
    p = _lib.Plan(n, 1, _lib.C64); p.set_field(x)
    X = p.debug_fft(); ref = np.fft.fft(x.astype(np.complex128), axis=-1)
    e1 = np.linalg.norm(X - ref) / np.linalg.norm(ref)
    e1n = np.linalg.norm(np.fft.fft(x, axis=-1) - ref) / np.linalg.norm(ref)
    p.set_field(x); H = np.ones(n, np.complex64)
    p.apply_transfer(H)
    y = p.get_field()
    e2 = np.linalg.norm(y - x) / np.linalg.norm(x)
    e2n = np.linalg.norm(np.fft.ifft(np.fft.fft(x, axis=-1), axis=-1) - x) / np.linalg.norm(x)
    # 100 round trips: the growth law
    p.set_field(x)
    for _ in range(100): p.apply_transfer(H)
    y = p.get_field(); e3 = np.linalg.norm(y - x) / np.linalg.norm(x)
    yn = x.copy()
    for _ in range(100): yn = np.fft.ifft(np.fft.fft(yn, axis=-1), axis=-1)
    e3n = np.linalg.norm(yn - x) / np.linalg.norm(x)
    say(f"n=2^{k}: fft {e1:.2e} | {e1n:.2e}   round trip {e2:.2e} | {e2n:.2e}   100 round trips {e3:.2e} | {e3n:.2e}")
    p.close()

say("# FIBER, SMF, h = 0.5 km, 4 mW QPSK-like field: max|A - A_float64|/peak, HIP | oracle (numpy complex64), and HIP against the oracle")
gv(**workloads.BENCH_GV)
for n, npol, steps in ((4096, 2, 100), (8192, 2, 100), (16384, 1, 100), (65536, 1, 100), (8192, 2, 30), (8176, 2, 100), (8176, 2, 30), (15060, 2, 100), (32752, 1, 100), (32752, 1, 30)):
    a = workloads.qpsk_field(1 << (n - 1).bit_length(), seed=n % 997, power_w=4e-3, n_pol=2)[:npol, :n]
    kw = dict(length=0.5 * steps, h=0.5, **workloads.SMF)
    y = oa.FIBER(optical_signal(a), **kw)
    o = orc.fiber_c64(a, gv.dt, **kw); t = S.run_f64(a, gv.dt, step_schedule(kw['length'], kw['h'])[0], workloads.SMF)      # (float64 arithmetic on the float32 coefficients and schedule)
    pk = np.max(np.abs(t))
    say(f"{n:6d} x {npol} {steps:4d} steps  engine {getattr(y, 'engine', '?'):24s} HIP {np.max(np.abs(y.signal - t)) / pk:.2e} | oracle {np.max(np.abs(o - t)) / pk:.2e}   HIP-oracle {np.max(np.abs(y.signal - o)) / pk:.2e}")
say("# 1000 steps (h = 0.125 km): energy against fl32(e^(-alpha h / 2))^(2 x 1000), and the distance from the oracle's strided fixtures where there is one")
for n, npol in ((8192, 2), (1 << 16, 1), (1 << 20, 2)):
    a = workloads.qpsk_field(n, seed=2024, n_pol=2)[:npol]
    kw = dict(length=125.0, h=0.125, **workloads.SMF)
    y = oa.FIBER(optical_signal(a), **kw).signal
    amp = np.float64(np.exp(np.float32(np.float32(-0.2 / 4.343) / 2) * np.float32(0.125)).astype(np.float32))
    e_in = np.sum(np.abs(a.astype(np.complex64).astype(np.complex128)) ** 2); e_out = np.sum(np.abs(y.astype(np.complex128)) ** 2)
    line = f"{n:8d} x {npol}: energy / expected - 1 = {e_out / (e_in * amp ** 2000) - 1:+.2e}"
    if n <= 8192:
        o = orc.fiber_c64(a, gv.dt, **kw)
        line += f"   HIP-oracle {np.max(np.abs(y - o)) / np.max(np.abs(o)):.2e}"
    say(line)
open(os.path.join(ROOT, "gpurun_out", os.environ.get("FFT_NOISE_OUT", "r05_fft_noise.txt")), "w").write("\n".join(out) + "\n")
