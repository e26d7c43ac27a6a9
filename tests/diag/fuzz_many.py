"""Stress run: many random FIBER / DBP configurations (power-of-two and other lengths, fixed and adaptive, both polarisation counts, loss and gain) against the
oracle AND the float64 solution (fuzz_cases.judge has the stated bound).  Prints the worst cases with all three distances; exits non-zero on a violation.
    python tests/diag/fuzz_many.py [count] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_cases as fc
import opticomlib_amd as oa
from opticomlib_amd import workloads
from opticomlib_amd.typing import gv, optical_signal

count = int(sys.argv[1]) if len(sys.argv) > 1 else 150
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 99
gv(**workloads.BENCH_GV)
rows, bad, beyond_plain = [], 0, 0
for i, n, npol, kw, a, pow2 in fc.cases(count, seed, pow2_only=os.environ.get("FUZZ_POW2_ONLY") == "1", minlog2=int(os.environ.get("FUZZ_MINLOG2", "8")),
                                        maxlog2=int(os.environ.get("FUZZ_MAXLOG2", "14"))):
    engine, steps, (ok, e_ho, e_ht, e_ot) = fc.run_case(oa, gv, optical_signal, kw, a)
    tol = fc.tol_of(steps)
    rows.append((e_ho / tol, e_ho, e_ht, e_ot, steps, n, npol, engine, ok, kw))
    bad += not ok
    beyond_plain += e_ho > tol
rows.sort(key=lambda w: -w[0])
print("# worst by HIP-oracle / tol: HIP-oracle, HIP-float64, oracle-float64 (max|d|/peak), steps, n x pol, engine")
for w in rows[:8]:
    print(f"HIP-oracle/tol {w[0]:.2f}  {w[1]:.2e} {w[2]:.2e} {w[3]:.2e}  steps {w[4]}  n {w[5]} x {w[6]}  {w[7]}{'' if w[8] else '  VIOLATION'}  {w[9]}")
print("# worst by HIP-float64 / oracle-float64 among the cases beyond half the tolerance from the float64 solution")
for w in sorted([w for w in rows if w[2] > 0.5 * fc.tol_of(w[4])], key=lambda w: -w[2] / max(w[3], 1e-30))[:5]:
    print(f"ratio {w[2] / max(w[3], 1e-30):.2f}  {w[1]:.2e} {w[2]:.2e} {w[3]:.2e}  steps {w[4]}  n {w[5]} x {w[6]}  {w[7]}{'' if w[8] else '  VIOLATION'}")
print(f"{count} cases (seed {seed}): {bad} beyond the stated bound; {beyond_plain} with HIP-oracle > the plain tolerance (the oracle's own distance from float64 is what carries those)")
sys.exit(1 if bad else 0)
