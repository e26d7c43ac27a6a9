"""Per-workgroup view of a few launches of a fixed-step run from a -DSSFM_TRACE=1 build: which workgroups of a launch end late, and where
they ran (XCD, CU).   SSFM_LIB=build/var/_ssfm_trace.so python tools/trace_wg.py"""
import os, sys, csv, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SSFM_TRACE_FILE"] = "/tmp/ssfm_trace.csv"
os.environ["SSFM_TRACE_RAW"] = "/tmp/ssfm_trace_raw.csv"
from opticomlib_amd import _lib, devices, workloads
n = 1 << 20; dt = 1.0 / (16 * 32e9)
hs, _ = devices.step_schedule(12.5, 0.125)          # 100 steps
a = workloads.qpsk_field(n, seed=1).astype(np.complex64)
p = _lib.Plan(n, 2, _lib.C64); p.set_linear_operator(devices.linear_operator(n, dt, 0.2, -21.7, 0.13)); p.set_field(a)
for _ in range(2):
    p.propagate_fixed(1.3, hs); p.synchronize()
rows = list(csv.DictReader(open("/tmp/ssfm_trace_raw.csv")))
by = collections.defaultdict(list)
for r in rows:
    by[int(r["slot"])].append(r)
t_all = min(int(r["start"]) for r in rows if int(r["start"]))
for slot in sorted(by)[:8]:
    R = [r for r in by[slot] if int(r["start"])]
    st = np.array([int(r["start"]) for r in R]) / 100.0; en = np.array([int(r["end"]) for r in R]) / 100.0
    xcc = np.array([int(r["xcc"]) for r in R]); hw = np.array([int(r["hw_id"]) for r in R]); blk = np.array([int(r["block"]) for r in R])
    cu = (hw >> 8) & 15; se = (hw >> 13) & 7; sh = (hw >> 12) & 1
    life = en - st
    kind = "TF"[int(R[0]["kind"])]; lane = R[0]["lane"]
    print(f"slot {slot} lane {lane} {kind}: start {st.min()-t_all/100:.2f} .. {st.max()-t_all/100:.2f}, end {en.min()-t_all/100:.2f} .. {en.max()-t_all/100:.2f}; life median {np.median(life):.2f} p10 {np.percentile(life,10):.2f} p90 {np.percentile(life,90):.2f} max {life.max():.2f}")
    print("   life by XCD:", " ".join(f"{x}:{np.median(life[xcc==x]):.2f}/{life[xcc==x].max():.2f}({(xcc==x).sum()})" for x in sorted(set(xcc))))
    print("   life by block id mod 8:", " ".join(f"{np.median(life[blk%8==x]):.2f}" for x in range(8)))
    order = np.argsort(en)
    print("   last 12 to end: " + " ".join(f"b{blk[i]}x{xcc[i]}se{se[i]}cu{cu[i]}:{life[i]:.1f}" for i in order[-12:]))
    # (XCD, SE, SH, CU) collisions inside this launch: two workgroups of ONE launch on one CU
    key = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    print(f"   distinct CUs {len(key)}; CUs holding 2+ workgroups of this launch: {sum(1 for v in key.values() if v > 1)}")
    # late starters
    late = st - st.min()
    print(f"   start lateness: p50 {np.median(late):.2f} p90 {np.percentile(late,90):.2f} max {late.max():.2f}; corr(life, start) {np.corrcoef(life, late)[0,1]:.2f}")
