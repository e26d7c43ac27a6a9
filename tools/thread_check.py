"""Host threads driving the library at once -- on different shapes (separate plans) AND on the same shapes (the same
cached plan: the device functions hold the plan's lock for their whole sequence): results must equal the single-threaded ones."""
import sys, threading; sys.path.insert(0, '.')
import numpy as np
import opticomlib_amd as oa
from opticomlib_amd import workloads
from opticomlib_amd.typing import gv, optical_signal
gv(**workloads.BENCH_GV)
cases = [(1 << 14, 2, 11), (1 << 15, 1, 12), (3000, 2, 13), (1 << 16, 2, 14)]
def run(case):
    n, npol, seed = case
    a = workloads.qpsk_field(1 << max(8, (n - 1).bit_length()), seed=seed, n_pol=2)[:npol, :n]
    a = a[0] if npol == 1 else a
    y = oa.FIBER(optical_signal(a), length=20, h=1.0, **workloads.SMF)
    v = oa.PD(oa.BPF(y, 200e9), BW=40e9, include_noise="none")
    return y.signal, v.signal
want = [run(c) for c in cases]
got = [None] * len(cases)
errs = []
def worker(idx):
    try:
        for _ in range(10):
            for i in idx:
                got[i] = run(cases[i])
    except Exception as e:       # noqa: BLE001
        errs.append(repr(e))
ts = [threading.Thread(target=worker, args=(idx,)) for idx in ([0, 2], [1, 3], [0, 1, 2, 3], [3, 2, 1, 0])]
[t.start() for t in ts]; [t.join() for t in ts]
def same(g, w, n):
    if n & (n - 1) == 0:
        return np.array_equal(g[0], w[0]) and np.array_equal(g[1], w[1])
    # a length that is not a power of two: the one-launch complex64 line, or -- should its workgroups not meet beside the other threads' launches -- the
    # general path's complex128 line: both within the tolerance of the reference's result, not bit for bit the same
    rel = lambda p, q: float(np.max(np.abs(np.asarray(p) - np.asarray(q))) / np.max(np.abs(np.asarray(q))))
    return rel(g[0], w[0]) < 2e-5 and rel(g[1], w[1]) < 1e-4
ok = not errs and all(same(g, w, c[0]) for g, w, c in zip(got, want, cases))
print("threads:", "identical to the single-threaded results" if ok else f"MISMATCH / errors: {errs}")
sys.exit(0 if ok else 1)
