"""Launch-level timeline of a fixed-step run from a -DSSFM_TRACE=1 build (dev aid).
usage: SSFM_LIB=build/abl/_ssfm_trace.so SSFM_GRAPH=0 python tools/trace_timeline.py"""
import os, sys, csv
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SSFM_TRACE_FILE"] = "/tmp/ssfm_trace.csv"
from opticomlib_amd import _lib, devices, workloads
n = 1 << 20; dt = 1.0 / (16 * 32e9)
hs, _ = devices.step_schedule(12.5, 0.125)          # 100 steps
a = workloads.qpsk_field(n, seed=1).astype(np.complex64)
p = _lib.Plan(n, 2, _lib.C64); p.set_linear_operator(devices.linear_operator(n, dt, 0.2, -21.7, 0.13)); p.set_field(a)
for _ in range(2):
    p.propagate_fixed(1.3, hs); p.synchronize()
rows = list(csv.DictReader(open("/tmp/ssfm_trace.csv")))
t0 = min(int(r["first_start"]) for r in rows)
ev = [(int(r["first_start"]) - t0, int(r["last_start"]) - t0, int(r["first_end"]) - t0, int(r["last_end"]) - t0, "TF"[int(r["kind"])], int(r["lane"])) for r in rows]
ev.sort()
print("times in us (100 MHz realtime counter); fs/ls = first/last workgroup start, fe/le = first/last workgroup end")
for e in ev[200:232]:
    print(f"lane {e[5]} {e[4]}: fs {e[0]/100:8.2f}  ls {e[1]/100:8.2f}  fe {e[2]/100:8.2f}  le {e[3]/100:8.2f}   dur {(e[3]-e[0])/100:6.2f}  ramp {(e[1]-e[0])/100:5.2f}")
# per-lane gaps between dependent launches
for lane in sorted(set(e[5] for e in ev)):
    L = [e for e in ev if e[5] == lane]
    gaps = [(L[i + 1][0] - L[i][3]) / 100 for i in range(50, len(L) - 1)]
    durs = [(e[3] - e[0]) / 100 for e in L[50:]]
    print(f"lane {lane}: median launch gap (last end -> next first start) {np.median(gaps):.2f} us, median kernel span {np.median(durs):.2f} us")
    for kind in "TF":
        K = [e for e in L[50:] if e[4] == kind]
        if not K:
            continue
        med = lambda f: float(np.median([f(e) for e in K])) / 100
        print(f"   {kind}: span {med(lambda e: e[3] - e[0]):.2f}  start spread {med(lambda e: e[1] - e[0]):.2f}  first workgroup done after {med(lambda e: e[2] - e[0]):.2f}"
              f"  end spread {med(lambda e: e[3] - e[2]):.2f} us")
total = (max(e[3] for e in ev) - min(e[0] for e in ev)) / 100
print(f"total {total:.1f} us for {len(hs)} steps -> {total/len(hs):.2f} us/step")
