"""LPF / BPF at BASELINE size for a rocprofv3 kernel trace (dev aid):
    rocprofv3 --kernel-trace --stats -d gpurun_out/sos_prof -- python3 tools/sos_prof.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from opticomlib_amd import workloads, gv, optical_signal, LPF, BPF, _lib
n = 1 << 20
gv(**workloads.BENCH_GV)
a = workloads.qpsk_field(n, seed=2024).astype(np.complex128)
x = optical_signal(a)
pw = np.abs(a[0]) ** 2
for name, f in (("BPF 2^20 x 2 complex128", lambda: BPF(x, BW=60e9)), ("LPF 2^20 float64", lambda: LPF(pw, BW=20e9))):
    f().signal
    t = time.perf_counter()
    for _ in range(3):
        f()
    left = (time.perf_counter() - t) / 3
    t = time.perf_counter()
    for _ in range(3):
        f().signal
    print(f"{name}: {(time.perf_counter() - t) / 3 * 1e3:.2f} ms host array to host array ({left * 1e3:.2f} ms with the result left in HBM), kernels {_lib.sosfiltfilt_last_ms() * 1e3:.1f} us")

# device-resident, back to back (clocks ramped): the kernels alone
from scipy import signal as sg
sos = sg.bessel(4, 30e9, "low", fs=gv.fs, norm="mag", output="sos"); zi = sg.sosfilt_zi(sos)
p, q = _lib.Plan(n, 2, _lib.C128), _lib.Plan(n, 2, _lib.C128)
p.set_field(a); p.synchronize()
for reps in (1, 20, 200):
    t = time.perf_counter()
    for _ in range(reps):
        _lib.sosfiltfilt_device(sos, zi, p.field_device_ptr, q.field_device_ptr, n, 2, True)
    w = (time.perf_counter() - t) / reps
    print(f"device-resident BPF x{reps}: wall {w * 1e6:.1f} us per call, kernels (last) {_lib.sosfiltfilt_last_ms() * 1e3:.1f} us")
