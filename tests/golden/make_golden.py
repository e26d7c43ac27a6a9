#!/usr/bin/env python3
"""Generate the golden fixtures by IMPORTING the reference (this container only).

    python tests/golden/make_golden.py [--reference /root/reference]

The reference's Python never travels to the GPU box; only the ``.npz`` outputs written
here do.  ``import opticomlib`` needs ``pympler`` (used only by ``.sizeof`` properties),
which is not installed: a two-line stand-in is created in a temp dir.
Recorded alongside every output: NumPy/SciPy versions.
"""
from __future__ import annotations

import argparse
import os
import sys
import tempfile

os.environ.setdefault("MPLBACKEND", "Agg")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402

from cases import CASES, case_drive, case_input  # noqa: E402


def import_reference(path):
    stub = tempfile.mkdtemp(prefix="pympler_stub_")
    os.makedirs(os.path.join(stub, "pympler"))
    open(os.path.join(stub, "pympler", "__init__.py"), "w").close()
    with open(os.path.join(stub, "pympler", "asizeof.py"), "w") as f:
        f.write("import sys\ndef asizeof(o):\n    return sys.getsizeof(o)\n")
    sys.path[:0] = [stub, path]
    import opticomlib  # noqa: F401
    from opticomlib import devices, typing
    return devices, typing


def run_case(name, case, devices, typing):
    typing.gv(**case["gv"])
    sig, noi = case_input(case)
    kw = dict(case["kw"])
    func = case["func"]
    if func == "PRBS":
        seq, last = devices.PRBS(return_seed=True, **kw)
        return {"out": np.asarray(seq.data, dtype=np.uint8), "seed_out": np.array(last, dtype=np.int64)}
    if func == "DAC":
        seq = devices.PRBS(order=case["bits"][0], len=case["bits"][1])
        kw2 = dict(kw)
        if "h" in kw2:
            kw2["h"] = np.asarray(kw2["h"], dtype=float)
        y = devices.DAC(seq, **kw2)
        assert y.noise is typing.NULL
        return {"out": y.signal, "bits": np.asarray(seq.data, dtype=np.uint8)}
    if func == "LASER":
        if "np_seed" in case:
            np.random.seed(case["np_seed"])
        y = devices.LASER(**kw)
        assert y.noise is typing.NULL
        return {"out": y.signal}
    if func == "MZM":
        v, vn = case_drive(case)
        el = v if vn is None else typing.electrical_signal(v, vn)
        x = typing.optical_signal(sig) if noi is None else typing.optical_signal(sig, noi)
        y = devices.MZM(x, el, **kw)
        out = {"out": y.signal}
        if y.noise is not typing.NULL:
            out["out_noise"] = y.noise
        return out
    if func == "LPF":
        x = None
    elif noi is None:
        x = typing.optical_signal(sig)
    else:
        x = typing.optical_signal(sig, noi)
    out = {}
    if func in ("FIBER", "DBP"):
        f = getattr(devices, func)
        if kw.get("return_steps"):
            z, A_z = f(x, **kw)
            out["z"], out["A_z"] = z, A_z
        else:
            y = f(x, **kw)
            assert y.noise is typing.NULL
            out["out"] = y.signal
            if case.get("want_z"):
                z, _ = f(x, return_steps=True, **kw)
                out["z"] = z
    elif func == "FIBER+DBP":
        y = devices.FIBER(x, **kw)
        out["mid"] = y.signal
        y2 = devices.DBP(y, **kw)
        out["out"] = y2.signal
    elif func == "DM":
        r = devices.DM(x, **kw)
        if kw.get("retH"):
            y, H = r
            out["H"] = H
        else:
            y = r
        out["out"] = y.signal
        if y.noise is not typing.NULL:
            out["out_noise"] = y.noise
    elif func == "LPF":
        xe = typing.electrical_signal(sig) if noi is None else typing.electrical_signal(sig, noi)
        r = devices.LPF(xe, **kw)
        if kw.get("retH"):
            y, H = r
            out["H"] = H
        else:
            y = r
        out["out"] = y.signal
        if y.noise is not typing.NULL:
            out["out_noise"] = y.noise
    elif func in ("PD", "EDFA"):
        if "np_seed" in case:
            np.random.seed(case["np_seed"])
        y = getattr(devices, func)(x, **kw)
        out["out"] = y.signal
        if y.noise is not typing.NULL:
            out["out_noise"] = y.noise
    elif func == "BPF":
        y = devices.BPF(x, **kw)
        out["out"] = y.signal
        if y.noise is not typing.NULL:
            out["out_noise"] = y.noise
    elif func == "TWIN":
        z, A_z, _, _ = devices.animated_fiber_propagation_with_phase(x, **kw)
        out["z"] = z
        out["A_last"] = A_z[-1]          # = A * exp(alpha*z/2)  (devices.py:2472)
    else:
        raise ValueError(func)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    devices, typing = import_reference(args.reference)
    import scipy
    meta = np.array([np.__version__, scipy.__version__])
    for name, case in CASES.items():
        if args.only and args.only != name:
            continue
        out = run_case(name, case, devices, typing)
        path = os.path.join(HERE, name + ".npz")
        np.savez(path, _versions=meta, **out)
        desc = ", ".join(f"{k}{tuple(np.shape(v))}:{np.asarray(v).dtype}" for k, v in out.items())
        print(f"{name:28s} {os.path.getsize(path)/1024:7.1f} KiB  {desc}")


if __name__ == "__main__":
    main()
