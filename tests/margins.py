"""One helper for every oracle / golden / fixture comparison of the GPU suite: measure max|d|/peak, hold it to the stated bound
(opticomlib_amd.accuracy.tol(steps) unless the test states another one), and append `test, what, steps, measured, bound, measured/bound`
to a file the suite writes (committed per round as profiles/rNN_parity_margins.txt) -- so the evidence says how far inside the
bound every green assertion is.  Test infrastructure."""
import os
import threading

import numpy as np

from opticomlib_amd import _lib
from opticomlib_amd.accuracy import TOL_C128, tol  # noqa: F401  (re-exported for the tests)
from opticomlib_amd.devices import step_schedule

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LOCK = threading.Lock()
_HEADER_DONE = set()


def margins_path():
    p = os.environ.get("SSFM_MARGINS_FILE")
    if p is None:
        p = os.path.join(ROOT, "gpurun_out", "parity_margins.txt")
    return p


def relmax(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / np.max(np.abs(b)))


def steps_of(kw):
    """Step count of a fixed-h run from its keyword arguments (the reference's float32 schedule); None for adaptive runs."""
    if kw is None or kw.get("h") is None:
        return None
    return len(step_schedule(kw["length"], kw["h"], _lib.C64)[0])


def _test_id():
    t = os.environ.get("PYTEST_CURRENT_TEST", "-")
    return t.split(" (")[0].replace(" ", "_")


def record(what, steps, measured, bound):
    path = margins_path()
    if not path:
        return
    try:
        with _LOCK:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "a") as f:
                if path not in _HEADER_DONE and f.tell() == 0:
                    f.write("# test | what | steps | measured max|d|/peak | bound | measured/bound\n")
                _HEADER_DONE.add(path)
                s = "-" if steps is None else str(int(steps))
                f.write(f"{_test_id()} | {what or '-'} | {s} | {measured:.3e} | {bound:.3e} | {measured / bound:.3f}\n")
    except OSError:
        pass


def within(got, want, bound=None, *, steps=None, kw=None, what="", measured=None):
    """True when max|got - want| / max|want| < bound; the bound defaults to tol(steps) with steps given or taken from a fixed-h `kw`.
    Every call is recorded with the error it measured."""
    if steps is None:
        steps = steps_of(kw)
    if bound is None:
        if steps is None:
            raise ValueError("within(): state a bound or the number of steps")
        bound = tol(steps)
    err = relmax(got, want) if measured is None else float(measured)
    record(what, steps, err, bound)
    if os.environ.get("SSFM_MARGINS_ONLY") == "1":      # (a survey run: record every margin, fail nothing)
        return True
    return err < bound
