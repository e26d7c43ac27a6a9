"""The stated accuracy contract of the complex64 split-step path, in ONE place.

SURVEY.md 8(c) states two points for a complex64 run against the reference's own complex64 run on the same
inputs: max|d|/peak <= 2e-5 after 100 steps and <= 3e-4 after 1000 steps.  Between and beyond them the bound is
continuous in the number of steps:

    tol(steps) = 2e-5                                   steps <= 100
                 2e-5 * (steps / 100) ** log10(15)      100 < steps <= 1000   (log-log line through the two points)
                 3e-4 * (steps / 1000)                  steps > 1000          (the systematic part grows with the steps)

Used by tests/test_gpu_parity.py (every comparison with the CPU restatement, the golden vectors and the fixtures), tests/diag/fuzz_cases.py and by the
routing rule of the one-launch complex64 chirp-z line (`c64_line_has_margin`, devices._fiber_chirpz_medium_c64).
Rounds 1-5 used a step function (2e-5 up to 100 steps, 3e-4 from step 101); nothing in this repository does any more.
"""
from __future__ import annotations

import math

TOL_AT_100 = 2e-5
TOL_AT_1000 = 3e-4
TOL_C128 = 1e-10
_EXPONENT = math.log10(TOL_AT_1000 / TOL_AT_100)          # 1.1761: (steps / 100) ** _EXPONENT is 15 at 1000 steps


def tol(steps) -> float:
    """max|d| / peak allowed between a complex64 run of `steps` steps and the reference's complex64 run of the same schedule."""
    s = float(steps)
    if s <= 100.0:
        return TOL_AT_100
    if s <= 1000.0:
        return TOL_AT_100 * (s / 100.0) ** _EXPONENT
    return TOL_AT_1000 * (s / 1000.0)


# The one-launch complex64 chirp-z line (lengths that are not powers of two, 2048 < n <= 65536): its own distance from the float64 solution.
# Measured over 242 random fibres of up to 100 steps (profiles/r05_chirp_margin.txt): <= 7.5e-7 x steps^0.75; re-measured in round 6 out to 1600 steps
# (tests/diag/chirp_line_law.py, profiles/r06_chirp_line_law.txt): the exponent holds, the envelope is 1.11 x higher (3000 x 2 at 800 / 1600 steps,
# 8176 x 2 at 50) -- the constant is 8.5e-7.  The reference's complex64 run is itself up to ~tol/2 ... tol from the float64 solution at such lengths,
# so the line may use HALF the contract bound; where its law exceeds that the run takes the complex128 line (1e-13 from float64).  With `tol` above:
# 8.5e-7 s^0.75 > tol(s) / 2 for 27 <= s <= 1031.
C64_LINE_LAW = (8.5e-7, 0.75)


def c64_line_error(steps) -> float:
    return C64_LINE_LAW[0] * float(max(steps, 0)) ** C64_LINE_LAW[1]


def c64_line_has_margin(steps) -> bool:
    """Whether a run of `steps` steps may take the one-launch complex64 chirp-z line: its measured error law within half the bound."""
    return c64_line_error(steps) <= 0.5 * tol(steps)


def c64_line_window() -> tuple:
    """(first, last) step count WITHOUT margin (runs inside take the complex128 line): derived from the law and `tol`, not fitted."""
    bad = [s for s in range(1, 20001) if not c64_line_has_margin(s)]
    return (bad[0], bad[-1]) if bad else (0, -1)
