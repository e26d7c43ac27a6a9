"""The oracle (oracle/ssfm_numpy.py) against outputs captured from the imported
reference (tests/golden/*.npz).  Bit-exact under the NumPy that wrote the fixtures;
under any other NumPy/CPU (different SIMD dispatch of complex multiply, libm) the
bound is a few float32 ulps of the field peak."""
import os
import warnings

import numpy as np
import pytest

from cases import CASES, case_drive, case_dt, case_input
from oracle import ssfm_numpy as orc


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _same_numpy(g):
    return str(g["_versions"][0]) == np.__version__


def _check(got, want, exact, rtol_peak):
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape
    assert got.dtype == want.dtype
    if exact:
        assert np.array_equal(got, want), f"max|d|={np.max(np.abs(got - want))}"
    else:
        assert np.max(np.abs(got - want)) <= rtol_peak * np.max(np.abs(want))


FIBER_CASES = [n for n, c in CASES.items() if c["func"] in ("FIBER", "DBP")]


@pytest.mark.parametrize("name", FIBER_CASES)
def test_fiber_dbp_matches_reference(golden_dir, name):
    case = CASES[name]
    g = _load(golden_dir, name)
    exact = _same_numpy(g)
    sig, noi = case_input(case)
    field = sig if noi is None else sig + noi
    f = orc.fiber_c64 if case["func"] == "FIBER" else orc.dbp_c64
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)      # gamma=0 adaptive divides by zero, like the reference
        if case["kw"].get("return_steps"):
            z, A_z = f(field, case_dt(case), **case["kw"])
            _check(z, g["z"], exact, 1e-6)
            _check(A_z, g["A_z"], exact, 2e-6)
            return
        out = f(field, case_dt(case), **case["kw"])
        _check(out, g["out"], exact, 2e-5)
        if "z" in g:
            z, _ = f(field, case_dt(case), return_steps=True, **case["kw"])
            _check(z, g["z"], exact, 1e-6)


def test_fiber_then_dbp(golden_dir):
    case = CASES["fiber_then_dbp"]
    g = _load(golden_dir, "fiber_then_dbp")
    exact = _same_numpy(g)
    sig, _ = case_input(case)
    mid = orc.fiber_c64(sig, case_dt(case), **case["kw"])
    _check(mid, g["mid"], exact, 2e-6)
    out = orc.dbp_c64(mid, case_dt(case), **case["kw"])
    _check(out, g["out"], exact, 2e-6)
    # KAT-3: the round trip is not the identity
    assert 1e-3 < np.max(np.abs(out - sig)) < 1e-2


@pytest.mark.parametrize("name", [n for n, c in CASES.items() if c["func"] == "DM"])
def test_dm_matches_reference(golden_dir, name):
    case = CASES[name]
    g = _load(golden_dir, name)
    exact = _same_numpy(g)
    sig, noi = case_input(case)
    out_s, out_n = orc.dm_c128(sig, case_dt(case), case["kw"]["D"], noise=noi)
    _check(out_s, g["out"], exact, 1e-13)
    if noi is not None:
        _check(out_n, g["out_noise"], exact, 1e-13)
    if case["kw"].get("retH"):
        H = np.fft.fftshift(orc.dm_transfer(sig.shape[-1], case_dt(case), case["kw"]["D"]))
        _check(H, g["H"], exact, 1e-13)


@pytest.mark.parametrize("name", [n for n, c in CASES.items() if c["func"] == "TWIN"])
def test_c128_twin(golden_dir, name):
    """float64 loop vs the reference's float64 twin (devices.py:2461-2486); the twin stores
    A*exp(alpha*z/2) (devices.py:2472)."""
    case = CASES[name]
    g = _load(golden_dir, name)
    sig, _ = case_input(case)
    kw = case["kw"]
    out = orc.fiber_c128(sig, case_dt(case), **kw)
    z_end = g["z"][-1]
    want = g["A_last"] * np.exp(-(kw["alpha"] / 4.343) * z_end / 2)
    assert abs(z_end - kw["length"]) < 1e-12
    assert np.max(np.abs(out - want)) <= 1e-14 * np.max(np.abs(want)) * 10


def test_known_answers_survey():
    """KAT-0 / KAT-2 literal values of SURVEY.md 8(c)."""
    case = CASES["kat0_fixed_2pol"]
    sig, _ = case_input(case)
    out = orc.fiber_c64(sig, case_dt(case), **case["kw"])
    assert abs(out[0, 0] - (-0.02527584 - 0.0087320795j)) < 2e-8
    assert abs(out[1, 1234] - (-0.05602411 + 0.03285698j)) < 2e-8
    dm, _ = orc.dm_c128(sig, case_dt(case), -200.0)
    assert abs(dm[0, 0] - (-0.03243257981209367 - 0.009760015724347008j)) < 1e-15
    assert abs(dm[1, 1234] - (-0.06793556274179213 + 0.04536301767925531j)) < 1e-15


def test_reference_own_tests_restated():
    """Reference tests/devices_test.py:257-277 on the oracle: attenuation (rtol 1e-3),
    DBP(FIBER(x)) with all-zero parameters is the identity (atol 1e-5)."""
    x = np.full(2048, np.sqrt(10e-3), dtype=np.complex128)            # LASER(P0=10 dBm) CW
    dt = 1.0 / 16e9
    y = orc.fiber_c64(x, dt, length=10, alpha=0.2)
    p_in = np.mean(np.abs(x) ** 2)
    np.testing.assert_allclose(np.mean(np.abs(y) ** 2), p_in * np.exp(-0.2 / 4.343 * 10), rtol=1e-3)
    w = orc.dbp_c64(orc.fiber_c64(x, dt, 10), dt, 10)
    np.testing.assert_allclose(w, x, atol=1e-5)


def test_step_schedule_float32():
    """z accumulates in float32 -- 100/0.1 is 1001 steps, 125/0.125 is 1000."""
    assert len(orc.step_schedule_c64(100, 0.1)) == 1001
    assert len(orc.step_schedule_c64(125, 0.125)) == 1000
    assert len(orc.step_schedule_c64(1000, 1.0)) == 1000
    s = orc.step_schedule_c64(2, 0.3)
    assert len(s) == 7 and s[-1] < s[0]


@pytest.mark.parametrize("name", [n for n, c in CASES.items() if c["func"] in ("LPF", "BPF")])
def test_bessel_filters_match_reference(golden_dir, name):
    """LPF / BPF: restated sosfiltfilt (oracle/filters_numpy.py) vs outputs of the imported reference."""
    from oracle import filters_numpy as fo
    case = CASES[name]
    g = _load(golden_dir, name)
    exact = _same_numpy(g)
    sig, noi = case_input(case)
    fs = case["gv"]["sps"] * case["gv"]["R"]
    kw = case["kw"]
    fn = fo.lpf if case["func"] == "LPF" else fo.bpf
    out, out_n = fn(sig, kw["BW"], fs, n=kw.get("n", 4), noise=noi)
    _check(out, g["out"], exact, 1e-13)
    if noi is not None:
        _check(out_n, g["out_noise"], exact, 1e-13)


@pytest.mark.parametrize("name", [n for n, c in CASES.items() if c["func"] in ("PD", "EDFA")])
def test_receiver_front_end_matches_reference(golden_dir, name):
    """PD / EDFA: restated front-end (oracle/frontend_numpy.py) vs outputs of the imported reference, the
    global np.random generator seeded as in tests/golden/make_golden.py."""
    from oracle import frontend_numpy as fe
    case = CASES[name]
    g = _load(golden_dir, name)
    exact = _same_numpy(g)
    sig, noi = case_input(case)
    fs = case["gv"]["sps"] * case["gv"]["R"]
    if "np_seed" in case:
        np.random.seed(case["np_seed"])
    if case["func"] == "PD":
        out, out_n = fe.pd(sig, noi, fs, **case["kw"])
    else:
        out, out_n = fe.edfa(sig, noi, fs, fe.default_f0(), **case["kw"])
    assert out.shape == g["out"].shape and out.dtype == g["out"].dtype
    _check(out, g["out"], exact, 1e-13)
    if "out_noise" in g:
        _check(out_n, g["out_noise"], exact, 1e-12)
    else:
        assert out_n is None


# reference tests/devices_test.py:52-71 (literal vectors of its own PRBS test)
REF_PRBS_20 = {
    7: [1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 1],
    9: [1, 0, 0, 0, 0, 0, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 0, 0, 0, 1],
    11: [1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0, 1],
    15: [1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0],
    20: [1, 0, 0, 0, 1, 1, 1, 0, 0, 0, 1, 1, 1, 0, 0, 0, 1, 1, 1, 0],
    23: [1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1],
    31: [1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0],
}


def test_prbs_oracle_against_the_references_own_test_vectors():
    from oracle import prbs_numpy as po
    for order, want in REF_PRBS_20.items():
        bits, _ = po.prbs(order, 20)
        assert bits.tolist() == want
    assert po.prbs(7, 10, seed=0)[0].tolist() == [1, 0, 0, 0, 0, 0, 1, 1, 0, 0]          # seed 0 -> 1 (devices_test.py:57)
    two, _ = po.prbs(7, 254)
    assert two.tolist() == po.prbs(7, 127)[0].tolist() * 2                                  # longer than one period (:71)


@pytest.mark.parametrize("name", [n for n, c in CASES.items() if c["func"] == "PRBS"])
def test_prbs_oracle_matches_reference(golden_dir, name):
    from oracle import prbs_numpy as po
    g = _load(golden_dir, name)
    kw = CASES[name]["kw"]
    bits, last = po.prbs(kw["order"], kw.get("len"), kw.get("seed"))
    np.testing.assert_array_equal(bits, g["out"])
    assert last == int(g["seed_out"])


def test_tidied_cpu_variant_is_bit_identical():
    """oracle.fiber_c64_tidy (the many-core CPU baseline of bench.py --cpu-manycore) computes the same values."""
    rng = np.random.default_rng(4)
    a = (rng.standard_normal((2, 2048)) + 1j * rng.standard_normal((2, 2048))) * 0.03
    kw = dict(length=3.3, alpha=0.2, beta_2=-21.7, beta_3=0.13, gamma=1.3, h=0.5)
    np.testing.assert_array_equal(orc.fiber_c64_tidy(a, 1.953125e-12, **kw), orc.fiber_c64(a, 1.953125e-12, **kw))


@pytest.mark.parametrize("name", [n for n, c in CASES.items() if c["func"] == "DAC"])
def test_dac_oracle_matches_reference(golden_dir, name):
    """DAC: restated pulses + upfir (oracle/transmitter_numpy.py) vs outputs of the imported reference."""
    from oracle import prbs_numpy as po, transmitter_numpy as tx
    case = CASES[name]
    g = _load(golden_dir, name)
    bits, _ = po.prbs(*case["bits"])
    np.testing.assert_array_equal(bits, g["bits"])
    kw = dict(case["kw"])
    if "h" in kw:
        kw["h"] = np.asarray(kw["h"], dtype=float)
    out = tx.dac(bits, case["gv"]["sps"], case["gv"]["sps"] * case["gv"]["R"], **kw)
    assert out.shape == g["out"].shape and out.dtype == g["out"].dtype
    _check(out, g["out"], _same_numpy(g), 1e-13)


@pytest.mark.parametrize("name", [n for n, c in CASES.items() if c["func"] in ("LASER", "MZM")])
def test_laser_mzm_oracle_matches_reference(golden_dir, name):
    from cases import case_drive
    from oracle import transmitter_numpy as tx
    case = CASES[name]
    g = _load(golden_dir, name)
    exact = _same_numpy(g)
    gvk = case["gv"]
    fs = gvk["sps"] * gvk["R"]
    if case["func"] == "LASER":
        if "np_seed" in case:
            np.random.seed(case["np_seed"])
        out = tx.laser(tx.time_vector(gvk["N"], gvk["sps"], fs), 1 / fs, fs, **case["kw"])
        assert out.dtype == g["out"].dtype
        _check(out, g["out"], exact, 1e-13)
        return
    sig, noi = case_input(case)
    v, vn = case_drive(case)
    out, out_n = tx.mzm(sig, noi, v, vn, fs, **case["kw"])
    _check(out, g["out"], exact, 1e-13)
    if "out_noise" in g:
        _check(out_n, g["out_noise"], exact, 1e-13)
    else:
        assert out_n is None
