"""CPU-only tests: the C-ABI library loads and exports what include/ssfm_amd.h declares, the
host-side mirror (types, coefficients, step schedule, error behaviour) matches the reference /
the oracle.  No compute call is made here (there is no GPU)."""
import os
import re

import numpy as np
import pytest

import opticomlib_amd as oa
from opticomlib_amd import _lib, devices
from opticomlib_amd.typing import NULL, gv, optical_signal
from oracle import ssfm_numpy as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ssfm_amd.h")).read()
    declared = set(re.findall(r"\b(ssfm_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"ssfm_plan"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ssfm_abi_version() == 3
    assert _lib.supported_log2n(_lib.C64) == (8, 24) and _lib.supported_log2n(_lib.C64, direct=True) == (8, 22)


def test_no_cpu_fallback_without_device():
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(oa.SsfmError):
        oa.Plan(4096, 2)
    gv(sps=16, R=10e9)
    with pytest.raises(oa.SsfmError):
        oa.FIBER(optical_signal(np.ones(4096, complex)), length=1, h=1.0)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "opticomlib_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src or f == "_never_", f"{f} mentions the oracle"


def test_type_errors_match_reference():
    # reference devices.py:1121-1122, 1021-1022: checked before anything touches the GPU
    for fn, kw in ((oa.FIBER, dict(length=1)), (oa.DBP, dict(length=1)), (oa.DM, dict(D=1.0))):
        with pytest.raises(TypeError, match="`input` must be of type 'optical_signal'."):
            fn(np.ones(16), **kw)


def test_unsupported_size_raises_value_error():
    """Sizes are validated before anything touches the device: 2 ... 2^21 samples (chirp-z path for lengths that
    are not powers of two), powers of two up to 2^22."""
    for n in ((1 << 21) + 1, (1 << 22) + 2, 3 << 20):
        with pytest.raises(ValueError, match="samples per polarisation"):
            devices._check_size(n, _lib.C64)
    with pytest.raises(ValueError, match="samples per polarisation"):
        devices._check_size(1, _lib.C64)
    with pytest.raises(ValueError, match="samples per polarisation"):
        devices._check_size(1 << 25, _lib.C64)
    for n in (2, 3, 64, 255, 256, 3000, 100003, 1 << 21, 1 << 22, 1 << 23, 1 << 24):          # (2^23, 2^24: split plans, round 6)
        devices._check_size(n, _lib.C64)                            # accepted
    assert devices._is_fast_size(4096, _lib.C64) and not devices._is_fast_size(3000, _lib.C64) and not devices._is_fast_size(128, _lib.C64)


def test_bad_step_rejected():
    with pytest.raises(ValueError):
        devices.step_schedule(10, 0.0)
    with pytest.raises(ValueError):
        devices.step_schedule(10, -1.0)


@pytest.mark.parametrize("length,h", [(100, 0.1), (125, 0.125), (1000, 1.0), (2, 0.3), (10, 1.0), (5, 7.0), (0, 1.0), (-3, 1.0)])
def test_step_schedule_matches_oracle(length, h):
    hs, z = devices.step_schedule(length, h)
    want = orc.step_schedule_c64(length, h)
    assert hs.dtype == np.float32 and np.array_equal(hs, want)
    assert len(z) == len(hs) + 1 and z[0] == 0
    if len(hs):
        assert z[-1] >= np.float32(length)


@pytest.mark.parametrize("n,dt", [(4096, 6.25e-12), (1 << 14, 1.953125e-12), (3000, 6.25e-12)])
def test_linear_operator_bit_exact_with_oracle(n, dt):
    kw = dict(alpha=0.2, beta_2=-21.7, beta_3=0.13)
    d = devices.linear_operator(n, dt, precision=_lib.C64, **kw)
    assert d.dtype == np.complex64
    assert np.array_equal(d, orc.linear_operator_c64(n, dt, **kw))
    d2 = devices.linear_operator(n, dt, precision=_lib.C128, **kw)
    assert np.array_equal(d2, orc.linear_operator_c128(n, dt, **kw))


class TestOpticalSignal:
    """shape -> n_pol table of reference tests/typing_test.py:766-811 and :1244-1251."""

    def test_errors(self):
        with pytest.raises(TypeError):
            optical_signal()
        with pytest.raises(ValueError):
            optical_signal([0, 1, 2], [0, 1, 2, 3])
        with pytest.raises(ValueError):
            optical_signal([[1, 2, 3], [5, 6, 7], [8, 9, 10]])
        with pytest.raises(ValueError):
            optical_signal([[[1, 2, 3]]])
        with pytest.raises(ValueError):
            optical_signal([1, 2, 3], n_pol=3)

    @pytest.mark.parametrize("mk", [list, tuple, np.array])
    def test_one_pol(self, mk):
        x = optical_signal(mk(range(6)), mk([-i for i in range(6)]), n_pol=1)
        assert np.array_equal(x.signal, np.arange(6)) and np.array_equal(x.noise, -np.arange(6))
        assert x.n_pol == 1 and x.size == 6 and x.execution_time == 0
        y = optical_signal(mk(range(6)))
        assert y.noise is NULL and y.n_pol == 1 and y.size == 6

    def test_two_pol_forms(self):
        base = np.arange(6)
        for sig in (base, base[None, :], np.tile(base, (2, 1))):
            x = optical_signal(sig, -sig, n_pol=2)
            assert np.array_equal(x.signal, np.tile(base, (2, 1)))
            assert np.array_equal(x.noise, -np.tile(base, (2, 1)))
            assert x.n_pol == 2 and x.size == 6 and len(x) == 6
        # (1, N) silently becomes dual-pol (typing.py:2176-2181); (2, N) with n_pol=1 keeps row 0
        assert optical_signal(base[None, :]).n_pol == 2
        assert optical_signal(np.tile(base, (2, 1)), n_pol=1).signal.shape == (6,)
        s = optical_signal(3.0, n_pol=2)
        assert s.signal.shape == (2, 1)

    def test_w_and_to_numpy(self):
        gv(sps=8, R=2e9)
        for sig in (np.ones(64), np.ones((2, 64))):
            x = optical_signal(sig)
            assert np.array_equal(x.w(), 2 * np.pi * np.fft.fftfreq(64) * gv.fs) or np.allclose(x.w(), 2 * np.pi * np.fft.fftfreq(64) * gv.fs, rtol=1e-15)
            assert np.allclose(x.w(shift=True), np.fft.fftshift(x.w()))
            assert np.array_equal(x.w(), orc.angular_frequency(64, gv.dt))
        x = optical_signal(np.ones(8), 2 * np.ones(8))
        assert np.array_equal(x.to_numpy(), 3 * np.ones(8))
        assert np.array_equal(optical_signal(np.ones(8)).to_numpy(), np.ones(8))
        gv.default()
        assert gv.dt == 1 / 16e9

    def test_gv_precedence(self):
        gv(sps=16, R=10e9)
        assert gv.fs == 16 * 10e9 and gv.dt == 1 / (16 * 10e9)
        gv(R=1e9, fs=32e9)
        assert gv.sps == 32
        gv(fs=8e9)
        assert gv.R == 1e9 and gv.sps == 8
        gv.default()


def test_electrical_signal_and_filter_argument_errors():
    from opticomlib_amd.typing import electrical_signal
    x = electrical_signal([1, 2, 3], [0, 0, 1])
    assert x.size == 3 and x.ndim == 1 and np.array_equal(x.to_numpy(), [1, 2, 4])
    assert electrical_signal(2.0).signal.shape == (1,)
    with pytest.raises(ValueError):
        electrical_signal(np.ones((2, 3)))
    with pytest.raises(ValueError):
        electrical_signal([1, 2, 3], [1, 2])
    # checked before anything touches the GPU (reference devices.py:811-812, :1355-1356)
    with pytest.raises(TypeError, match=r"`input` must be of type \(optical_signal\)."):
        oa.BPF(np.ones(64), 1e9)


def test_missing_library_fails_loudly():
    """No silent fallback: with the shared library absent every entry point raises SsfmError."""
    import subprocess
    import sys
    code = (
        "import numpy as np, opticomlib_amd as oa\n"
        "from opticomlib_amd.typing import gv, optical_signal\n"
        "gv(sps=16, R=10e9)\n"
        "try:\n"
        "    oa.FIBER(optical_signal(np.ones(4096, complex)), length=1, h=1.0)\n"
        "except oa.SsfmError as e:\n"
        "    assert 'missing' in str(e) and 'no CPU fallback' in str(e), e\n"
        "    print('LOUD')\n")
    env = dict(os.environ, SSFM_LIB="/nonexistent/_ssfm_amd.so", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "LOUD" in r.stdout, r.stderr[-2000:]


def test_front_end_argument_errors_need_no_gpu():
    """PD / EDFA validate their arguments before touching the device (reference devices.py:1490-1510, :915-916)."""
    import opticomlib_amd as oa
    x = oa.optical_signal(np.ones(64, complex))
    with pytest.raises(TypeError, match="`input` must be of type 'optical_signal'."):
        oa.PD(np.ones(64), 1e9)
    with pytest.raises(TypeError, match="`input` must be of type 'optical_signal'."):
        oa.EDFA(np.ones(64), 20, 5)
    with pytest.raises(ValueError, match=r"`r` must be in the range \(0,1\]"):
        oa.PD(x, 1e9, r=1.5)
    with pytest.raises(TypeError, match="`r` must be a scalar value."):
        oa.PD(x, 1e9, r="1")
    with pytest.raises(ValueError, match="`T` must be a positive value."):
        oa.PD(x, 1e9, T=-1.0)
    with pytest.raises(ValueError, match="`R_load` must be a positive value."):
        oa.PD(x, 1e9, R_load=-50.0)
    with pytest.raises(TypeError, match="`include_noise` must be a string."):
        oa.PD(x, 1e9, include_noise=None)


def test_gv_optical_carrier():
    from opticomlib_amd.typing import gv
    gv(sps=8, R=1e9)
    assert gv.wavelength == 1550e-9 and abs(gv.f0 - 299792458.0 / 1550e-9) < 1
    gv(sps=8, R=1e9, wavelength=1310e-9)
    assert abs(gv.f0 - 299792458.0 / 1310e-9) < 1
    gv(sps=8, R=1e9)
    assert gv.wavelength == 1550e-9


def test_foreign_signal_objects_are_adopted():
    """devices._adopt: an object with the reference library's layout is converted, the sampling grid comes from
    ITS module's gv, and results go back as the caller's class (tests/foreign_types.py stands in for opticomlib)."""
    import foreign_types as ft
    from opticomlib_amd import devices as od
    from opticomlib_amd.typing import NULL, gv, optical_signal, electrical_signal
    gv(sps=8, R=1e9)
    ft.gv.set(16, 32e9)
    x = ft.optical_signal(np.ones((2, 64), complex), 0.1 * np.ones((2, 64), complex))
    mine, grid, back = od._adopt(x, "optical_signal")
    assert isinstance(mine, optical_signal) and mine.n_pol == 2 and grid is ft.gv and grid.fs == 512e9
    np.testing.assert_array_equal(mine.noise, x.noise)
    out = optical_signal(2 * mine.signal)
    out.execution_time = 1.5
    res = back(out)
    assert type(res) is ft.optical_signal and res.noise is ft.NULL and res.execution_time == 1.5
    res2, H = back((optical_signal(mine.signal, mine.noise), np.ones(3)))
    assert type(res2) is ft.optical_signal and isinstance(res2.noise, np.ndarray) and H.shape == (3,)
    e = back(electrical_signal(np.ones(64)))                   # PD: optical in, electrical out
    assert type(e) is ft.electrical_signal
    z = back((np.zeros(3), np.zeros((3, 2, 64))))              # return_steps tuple passes through
    assert isinstance(z[0], np.ndarray)
    # our own objects and non-signals pass through untouched
    own = optical_signal(np.ones(8, complex))
    assert od._adopt(own, "optical_signal")[0] is own and od._adopt(own, "optical_signal")[1] is gv
    assert od._adopt(np.ones(4), "optical_signal")[0].shape == (4,)
    y = ft.optical_signal(np.ones(64, complex))
    assert od._adopt(y, "optical_signal")[0].noise is NULL and od._adopt(y, "optical_signal")[0].n_pol == 1


# ------------------------------------------------------------------ PRBS: argument checks (the generator is HIP: tests/test_gpu_parity.py)
def test_prbs_argument_errors_come_before_any_device_work():
    import opticomlib_amd as oa
    with pytest.raises(TypeError, match="must be an integer"):
        oa.PRBS(order=15, len="20")
    with pytest.raises(ValueError, match="must be one of the following values"):
        oa.PRBS(order=8)
    with pytest.raises(ValueError, match="greater than cero"):
        oa.PRBS(order=7, len=0)
    if _lib.device_count() == 0:
        with pytest.raises(oa.SsfmError):                  # no CPU generator behind it
            oa.PRBS(7)


def test_binary_sequence_container():
    import opticomlib_amd as oa
    b = oa.binary_sequence("0110 1")
    assert b.size == 5 and len(b) == 5 and b.ones == 3 and b.zeros == 2 and b.type is oa.binary_sequence
    assert b.data.dtype == np.uint8 and b[1] == 1 and b[1:3].data.tolist() == [1, 1]
    assert np.all(b == [0, 1, 1, 0, 1]) and np.asarray(b).tolist() == [0, 1, 1, 0, 1]
    with pytest.raises(ValueError, match="only 0 and 1"):
        oa.binary_sequence([0, 2])
    with pytest.raises(ValueError, match="must be 1D"):
        oa.binary_sequence([[0, 1]])


# ------------------------------------------------------------------ LASER / MZM: argument checks (the arithmetic is HIP: tests/test_gpu_parity.py)
def test_laser_and_mzm_argument_checks_and_no_cpu_path():
    import opticomlib_amd as oa
    from opticomlib_amd import _lib
    from opticomlib_amd.typing import gv, optical_signal
    gv(sps=16, R=10e9, N=64)
    assert gv.t.size == 1024 and gv.N == 64
    with pytest.raises(TypeError, match="`op_input` must be of type 'optical_signal'."):
        oa.MZM(np.ones(8), 1.0)
    with pytest.raises(ValueError, match="`pol`"):
        oa.MZM(optical_signal(np.ones(8, complex)), 1.0, pol="z")
    with pytest.raises(ValueError, match="Nyquist"):
        oa.LASER(P0=0, df=1e12)
    if _lib.device_count() == 0:                                      # no GPU: no silent host arithmetic either
        with pytest.raises(_lib.SsfmError):
            oa.LASER(P0=0)
        with pytest.raises(_lib.SsfmError):
            oa.MZM(optical_signal(np.ones(8, complex)), np.ones(8))


def test_optical_signal_algebra():
    """The few operators a link script uses around the devices (reference typing.py:1308-1344, 1663-1720)."""
    from opticomlib_amd.typing import NULL, optical_signal
    rng = np.random.default_rng(0)
    s1, n1 = rng.standard_normal((2, 32)) + 0j, 0.1 * rng.standard_normal((2, 32)) + 0j
    x = optical_signal(s1, n1)
    y = optical_signal(2 * s1)
    p = x * y
    np.testing.assert_array_equal(p.signal, s1 * (2 * s1))
    np.testing.assert_array_equal(p.noise, n1 * (2 * s1))
    q = x * x.conj()
    np.testing.assert_array_equal(q.noise, s1 * n1.conj() + n1 * s1.conj() + n1 * n1.conj())
    assert (y * 3.0).noise is NULL and np.array_equal((3.0 * y).signal, 6 * s1)
    np.testing.assert_array_equal((x + y).signal, 3 * s1)
    np.testing.assert_array_equal((x + y).noise, n1)
    np.testing.assert_array_equal((x - y).signal, -s1)
    np.testing.assert_array_equal((-x).noise, -n1)
    np.testing.assert_allclose(x.power(), np.mean(np.abs(s1 + n1) ** 2, axis=-1))
    np.testing.assert_allclose(x.power("dBm", "signal"), 10 * np.log10(np.mean(np.abs(s1) ** 2, axis=-1)) + 30)
    assert not y.abs("noise").any() and x.abs("noise").shape == (2, 32)
    assert x[4:10].signal.shape == (2, 6) and optical_signal(s1[0])[::2].size == 16
    with pytest.raises(ValueError):
        x.power("mW")


def test_plan_labels_do_not_collide_like_python_hashes():
    """hash(-1.0) == hash(-2.0) in CPython: a hash-derived label made FIBER(beta_2=-2) reuse the operator staged for
    FIBER(beta_2=-1).  The label is a digest of the exact values."""
    from opticomlib_amd.devices import _tag
    assert hash(-1.0) == hash(-2.0)                                      # the trap itself
    seen = set()
    for a in (0.0, 0.2, -0.2, 1.0, -1.0, 2.0, -2.0):
        for b2 in (-1.0, -2.0, 1.0, 2.0, -21.7, 0.0):
            for b3 in (-1.0, -2.0, 0.0, 0.13):
                t = _tag("fibre", 1.953125e-12, a, b2, b3)
                assert t & 1 and 0 < t < 2 ** 64
                seen.add(t)
    assert len(seen) == 7 * 6 * 4
    assert _tag("chirp", 3000) == _tag("chirp", 3000) != _tag("chirp", 3001)
    assert _tag("gauss-shape", 4096, 16) != _tag("gauss-shape", 16, 4096)
    with pytest.raises(TypeError):
        _tag("x", [1, 2])


def test_the_stated_tolerance_is_one_continuous_bound_and_the_routing_window_follows_from_it():
    """opticomlib_amd.accuracy: SURVEY.md 8(c)'s two points (2e-5 @ 100 steps, 3e-4 @ 1000) joined by a log-log line, flat below, proportional
    beyond -- no step anywhere; the complex64 chirp-z line's window is derived from its measured error law against HALF of that bound."""
    from opticomlib_amd import accuracy as acc
    assert acc.tol(1) == acc.tol(100) == 2e-5 and abs(acc.tol(1000) - 3e-4) < 1e-18 and abs(acc.tol(2000) - 6e-4) < 1e-18
    s = np.arange(1, 5000)
    t = np.array([acc.tol(k) for k in s])
    assert np.all(np.diff(t) >= 0) and np.max(t[1:] / t[:-1]) < 1.013            # monotone, and no jump: at most the line's own slope per step
    assert abs(acc.tol(101) / acc.tol(100) - 1) < 0.012                            # (rounds 3-5: a factor 15 here)
    lo, hi = acc.c64_line_window()
    assert (lo, hi) == (27, 1031)
    assert acc.c64_line_has_margin(lo - 1) and not acc.c64_line_has_margin(lo) and not acc.c64_line_has_margin(hi) and acc.c64_line_has_margin(hi + 1)
    assert all(acc.c64_line_error(k) <= 0.5 * acc.tol(k) for k in (1, 10, 26, 1032, 2000, 5000))
    assert devices._c64_line_has_margin is acc.c64_line_has_margin and devices._C64_LINE_NO_MARGIN == (lo, hi)


def test_the_dynamic_symbol_table_is_the_c_abi_and_nothing_else():
    """-fvisibility=hidden + csrc/exports.map: `nm -D --defined-only` of the shared library lists the 59 prototypes of include/ssfm_amd.h, no C++
    symbol, no kernel handle (round 5: 535 of them)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = {line.split()[-1] for line in out.splitlines() if line.strip()}
    assert names == set(_lib.SYMBOLS), names ^ set(_lib.SYMBOLS)
    und = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "hipfft" not in und.lower() and "rocfft" not in und.lower() and "torch" not in und.lower()


def test_every_entry_point_the_documents_name_exists():
    """the header's own mapping table, DESIGN.md and INTEGRATION.md named entry points that the ABI fold of round 5 had removed.  Every
    full `ssfm_...` name in those texts is a declared prototype (or a type / constant of the header, or one of the names listed as REMOVED)."""
    hdr = open(os.path.join(ROOT, "include", "ssfm_amd.h")).read()
    known = set(_lib.SYMBOLS) | set(re.findall(r"\b(ssfm_[a-z0-9_]+)\b", re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)))
    removed_ok = re.compile(r"removed|hidden|round 4 exported|folded|no longer|internal|were|became|used to", re.I)
    for name in ("include/ssfm_amd.h", "INTEGRATION.md"):
        text = open(os.path.join(ROOT, name)).read()
        for m in re.finditer(r"\b(ssfm_[a-z0-9_]+)\b", text):
            w = m.group(1)
            if w in known or w.endswith("_") or w in ("ssfm_amd", "ssfm_host", "ssfm_kernels", "ssfm_common", "ssfm_medium", "ssfm_numpy", "ssfm_split"):
                continue
            line = text[text.rfind("\n", 0, m.start()) + 1: text.find("\n", m.end())]
            para = text[max(0, m.start() - 600): m.end() + 300]
            assert removed_ok.search(para), f"{name}: `{w}` is not an entry point of the ABI: {line.strip()[:160]}"
