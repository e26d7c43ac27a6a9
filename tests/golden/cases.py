"""Golden-vector case table shared by ``make_golden.py`` (which runs the imported
reference) and the tests (which run the oracle / the HIP path on the same inputs).

Inputs are regenerated from seeds (``np.random.default_rng`` = PCG64, stream stable
across NumPy versions), so the committed fixtures hold only the reference's OUTPUTS.
"""
from __future__ import annotations

import numpy as np


def make_input(kind: str, seed: int, shape, dtype="complex128", amp=0.03):
    """Seeded test field.

    ``noise``: white complex Gaussian (fills the whole simulated band -- stresses the
    band edge of the dispersion operator).  ``qpsk``: 16-samples-per-symbol QPSK-like
    pulses, Gaussian spectral shaping (band-limited, like the bench workload).
    """
    rng = np.random.default_rng(seed)
    shape = tuple(np.atleast_1d(shape))
    if kind == "noise":
        a = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) * amp
    elif kind == "qpsk":
        n = shape[-1]
        sps = 16
        nsym = n // sps
        b = rng.integers(0, 2, size=shape[:-1] + (nsym, 2))
        sym = ((2 * b[..., 0] - 1) + 1j * (2 * b[..., 1] - 1)) / np.sqrt(2)
        x = np.zeros(shape, dtype=np.complex128)
        x[..., sps // 2::sps] = sym
        f = np.fft.fftfreq(n) * sps                      # in units of the symbol rate
        x = np.fft.ifft(np.fft.fft(x, axis=-1) * np.exp(-(f / 0.6) ** 2 * np.log(2)), axis=-1)
        x *= amp / np.sqrt(np.mean(np.abs(x) ** 2, axis=-1, keepdims=True))
        a = x
    else:
        raise ValueError(kind)
    if np.dtype(dtype).kind == "f":
        a = a.real
    return np.ascontiguousarray(a.astype(dtype))


# gv(sps, R) -> dt = 1/(sps*R)
GV_A = dict(sps=16, R=10e9)       # dt = 6.25 ps
GV_B = dict(sps=16, R=32e9)       # dt = 1.953125 ps (bench grid)

GV_N = dict(sps=16, R=10e9, N=64)   # with a slot count: gv.t has 1024 samples

FIB = dict(alpha=0.2, beta_2=-20.0, beta_3=0.1, gamma=2.0)
SMF = dict(alpha=0.2, beta_2=-21.7, beta_3=0.13, gamma=1.3)

# name -> dict(func, gv, inp=(kind, seed, shape, dtype, amp), noise=(...)|None, kw)
CASES = {
    # KAT-0 of SURVEY.md 8(c)
    "kat0_fixed_2pol": dict(func="FIBER", gv=GV_A, inp=("noise", 0, (2, 4096), "complex128", 0.03),
                            kw=dict(length=10, h=1.0, **FIB)),
    # KAT-1: adaptive, phi_max default 0.01
    "kat1_adaptive_2pol": dict(func="FIBER", gv=GV_A, inp=("noise", 0, (2, 4096), "complex128", 0.03),
                               kw=dict(length=10, alpha=0.2, beta_2=-20.0, gamma=2.0), want_z=True),
    "adaptive_phi05_2pol_8k": dict(func="FIBER", gv=GV_B, inp=("qpsk", 5, (2, 8192), "complex128", 0.1),
                                   kw=dict(length=40, phi_max=0.05, **SMF), want_z=True),
    "fixed_short_last_1pol": dict(func="FIBER", gv=GV_A, inp=("noise", 1, (2048,), "complex128", 0.03),
                                  kw=dict(length=2, h=0.3, **FIB)),
    "fixed_1001_steps_1pol": dict(func="FIBER", gv=GV_A, inp=("qpsk", 2, (1024,), "complex128", 0.03),
                                  kw=dict(length=100, h=0.1, alpha=0.2, beta_2=-20.0, gamma=2.0), want_z=True),
    "fixed_alpha0_2pol": dict(func="FIBER", gv=GV_B, inp=("qpsk", 3, (2, 4096), "complex128", 0.0316),
                              kw=dict(length=20, h=0.5, alpha=0.0, beta_2=-21.7, beta_3=0.13, gamma=1.3)),
    "gamma0_single_step": dict(func="FIBER", gv=GV_A, inp=("noise", 4, (2, 4096), "complex128", 0.03),
                               kw=dict(length=10, alpha=0.2, beta_2=-20.0, beta_3=0.1, gamma=0.0)),
    "beta0_pure_spm": dict(func="FIBER", gv=GV_A, inp=("noise", 6, (2, 4096), "complex128", 0.03),
                           kw=dict(length=10, alpha=0.2, gamma=2.0)),
    "attenuation_only": dict(func="FIBER", gv=GV_A, inp=("noise", 7, (2048,), "complex128", 0.1),
                             kw=dict(length=10, alpha=0.2)),
    "dbp_negative_ops": dict(func="DBP", gv=GV_A, inp=("noise", 8, (2, 4096), "complex128", 0.03),
                             kw=dict(length=10, h=1.0, **FIB)),
    "with_noise_c64_input": dict(func="FIBER", gv=GV_A, inp=("qpsk", 9, (2, 4096), "complex64", 0.03),
                                 noise=("noise", 10, (2, 4096), "complex64", 0.003),
                                 kw=dict(length=5, h=0.5, **FIB)),
    "real_input_1pol": dict(func="FIBER", gv=GV_A, inp=("qpsk", 11, (4096,), "float64", 0.03),
                            kw=dict(length=5, h=1.0, **FIB)),
    "npow2_3000_1pol": dict(func="FIBER", gv=GV_A, inp=("noise", 12, (3000,), "complex128", 0.03),
                            kw=dict(length=3, h=1.0, **FIB)),
    "fixed_2pol_16k": dict(func="FIBER", gv=GV_B, inp=("qpsk", 13, (2, 16384), "complex128", 0.0316),
                           kw=dict(length=5, h=0.5, **SMF)),
    "fixed_2pol_16k_noiseband": dict(func="FIBER", gv=GV_B, inp=("noise", 14, (2, 16384), "complex128", 0.02),
                                     kw=dict(length=4, h=0.25, **SMF)),
    "small_256_1pol": dict(func="FIBER", gv=GV_A, inp=("noise", 15, (256,), "complex128", 0.03),
                           kw=dict(length=4, h=1.0, **FIB)),
    "small_512_2pol": dict(func="FIBER", gv=GV_A, inp=("noise", 16, (2, 512), "complex128", 0.03),
                           kw=dict(length=4, h=1.0, **FIB)),
    "return_steps_1k": dict(func="FIBER", gv=GV_A, inp=("noise", 17, (2, 1024), "complex128", 0.03),
                            kw=dict(length=5, h=1.0, return_steps=True, **FIB)),
    # KAT-3: DBP(FIBER(x)) -- the round trip is NOT the identity (stale N^)
    "fiber_then_dbp": dict(func="FIBER+DBP", gv=GV_A, inp=("noise", 0, (2, 4096), "complex128", 0.03),
                           kw=dict(length=10, h=1.0, **FIB)),
    # DM (complex128, signal and noise kept apart) -- KAT-2
    "dm_2pol": dict(func="DM", gv=GV_A, inp=("noise", 0, (2, 4096), "complex128", 0.03),
                    kw=dict(D=-200.0)),
    "dm_noise_retH": dict(func="DM", gv=GV_A, inp=("qpsk", 20, (2, 2048), "complex128", 0.03),
                          noise=("noise", 21, (2, 2048), "complex128", 0.003),
                          kw=dict(D=4000.0, retH=True)),
    "dm_1pol_8k": dict(func="DM", gv=GV_B, inp=("qpsk", 22, (8192,), "complex128", 0.03),
                       kw=dict(D=-21.7 * 80)),
    # zero-phase Bessel filters (reference devices.py:1286-1375 LPF, :788-826 BPF)
    "lpf_real_4k": dict(func="LPF", gv=GV_A, inp=("qpsk", 40, (4096,), "float64", 0.03),
                        kw=dict(BW=10e9)),
    "lpf_noise_n2_retH": dict(func="LPF", gv=GV_A, inp=("qpsk", 41, (2048,), "float64", 0.03),
                              noise=("noise", 42, (2048,), "float64", 0.003), kw=dict(BW=5e9, n=2, retH=True)),
    "lpf_complex_in_n6": dict(func="LPF", gv=GV_B, inp=("noise", 43, (8192,), "complex128", 0.03),
                              kw=dict(BW=40e9, n=6)),
    "bpf_2pol_4k": dict(func="BPF", gv=GV_A, inp=("noise", 44, (2, 4096), "complex128", 0.03),
                        noise=("noise", 45, (2, 4096), "complex128", 0.003), kw=dict(BW=40e9)),
    "bpf_1pol_8k": dict(func="BPF", gv=GV_B, inp=("qpsk", 46, (8192,), "complex128", 0.03),
                        kw=dict(BW=50e9, n=4)),
    "bpf_2pol_n3_3000": dict(func="BPF", gv=GV_B, inp=("noise", 47, (2, 3000), "complex128", 0.03),
                             kw=dict(BW=100e9, n=3)),
    # receiver front-end (reference devices.py:1378-1555 PD, :829-942 EDFA); the random terms come from the
    # global np.random generator, seeded with `np_seed` right before the call
    "pd_none_2pol": dict(func="PD", gv=GV_A, inp=("qpsk", 50, (2, 4096), "complex128", 0.03),
                         kw=dict(BW=8e9, include_noise="none")),
    "pd_ase_only_2pol": dict(func="PD", gv=GV_A, inp=("qpsk", 51, (2, 4096), "complex128", 0.03),
                             noise=("noise", 52, (2, 4096), "complex128", 0.003),
                             kw=dict(BW=8e9, r=0.8, R_load=75.0, include_noise="ase-only", i_dark=5e-9)),
    "pd_all_2pol": dict(func="PD", gv=GV_A, inp=("qpsk", 53, (2, 4096), "complex128", 0.03),
                        noise=("noise", 54, (2, 4096), "complex128", 0.003), np_seed=11,
                        kw=dict(BW=8e9, include_noise="all")),
    "pd_all_1pol_nonoise_in": dict(func="PD", gv=GV_B, inp=("qpsk", 55, (8192,), "complex128", 0.03), np_seed=12,
                                   kw=dict(BW=20e9, r=0.9, T=350.0, include_noise="ALL", Fn=3)),
    "pd_thermal_shot_1pol": dict(func="PD", gv=GV_A, inp=("noise", 56, (3000,), "complex128", 0.03),
                                 noise=("noise", 57, (3000,), "complex128", 0.003), np_seed=13,
                                 kw=dict(BW=5e9, include_noise="thermal-shot")),
    "edfa_2pol_bw": dict(func="EDFA", gv=GV_A, inp=("qpsk", 60, (2, 4096), "complex128", 0.003),
                         noise=("noise", 61, (2, 4096), "complex128", 0.0003), np_seed=21,
                         kw=dict(G=20, NF=5, BW=40e9)),
    "edfa_1pol_nobw": dict(func="EDFA", gv=GV_B, inp=("qpsk", 62, (8192,), "complex128", 0.003), np_seed=22,
                           kw=dict(G=17.5, NF=4.5)),
    "edfa_1pol_bw": dict(func="EDFA", gv=GV_B, inp=("qpsk", 63, (4096,), "complex128", 0.003), np_seed=23,
                         kw=dict(G=25, NF=6, BW=100e9)),
    # pseudo-random bit sequences (reference devices.py:63-182): integer LFSR, bit-exact; out = bits, seed_out = final state
    "prbs7_two_periods": dict(func="PRBS", gv=GV_A, kw=dict(order=7, len=254)),
    "prbs9_seed": dict(func="PRBS", gv=GV_A, kw=dict(order=9, len=1500, seed=124)),
    "prbs15_seed": dict(func="PRBS", gv=GV_A, kw=dict(order=15, len=70000, seed=12345)),
    "prbs20_default": dict(func="PRBS", gv=GV_A, kw=dict(order=20, len=5000)),
    "prbs23_seed_wraps": dict(func="PRBS", gv=GV_A, kw=dict(order=23, len=3000, seed=(1 << 23) + 77)),
    "prbs31_seed": dict(func="PRBS", gv=GV_A, kw=dict(order=31, len=4000, seed=0x5EED5EED)),
    "prbs11_short": dict(func="PRBS", gv=GV_A, kw=dict(order=11, len=5, seed=3)),
    # DAC pulse shaping (reference devices.py:185-350); input bits = PRBS(order, len) of the case, gv.sps from `gv`
    "dac_nrz_default": dict(func="DAC", gv=GV_A, bits=(7, 100), kw=dict()),
    "dac_nrz_T2_ac": dict(func="DAC", gv=GV_A, bits=(9, 64), kw=dict(pulse_shape="nrz", T=2, Vpp=2.5, offset=-1.0, coupling="AC")),
    "dac_gauss": dict(func="DAC", gv=GV_A, bits=(9, 128), kw=dict(pulse_shape="gaussian", Vpp=5.0, offset=-2.5)),
    "dac_gauss_chirp": dict(func="DAC", gv=GV_A, bits=(11, 77), kw=dict(pulse_shape="gaussian", T=2, c=0.7)),
    "dac_supergauss_m2": dict(func="DAC", gv=GV_A, bits=(11, 90), kw=dict(pulse_shape="gaussian", m=2, Vpp=3.0)),
    "dac_rcos_normal": dict(func="DAC", gv=GV_A, bits=(15, 200), kw=dict(pulse_shape="rcos", beta=0.25)),
    "dac_rcos_sqrt_bw": dict(func="DAC", gv=GV_A, bits=(15, 150), kw=dict(pulse_shape="rcos", beta=0.5, rcos_type="sqrt", BW=8e9)),
    "dac_custom_h": dict(func="DAC", gv=GV_A, bits=(7, 50), kw=dict(h=[0.25, 0.5, 1.0, 0.5, 0.25])),
    "dac_short": dict(func="DAC", gv=GV_A, bits=(7, 3), kw=dict(pulse_shape="gaussian")),
    # LASER (devices.py:353-510) and MZM (:620-786); `el` / `el_noise` = seeded drive voltage (and its noise)
    "laser_cw": dict(func="LASER", gv=GV_N, kw=dict(P0=10)),
    "laser_lw_rin_df": dict(func="LASER", gv=GV_N, np_seed=31, kw=dict(P0=5, lw=1e6, rin=-140, df=1e9)),
    "mzm_2pol_noise": dict(func="MZM", gv=GV_N, inp=("noise", 70, (2, 1024), "complex128", 0.03), noise=("noise", 71, (2, 1024), "complex128", 0.003),
                           el=(72, 2.0), el_noise=(73, 0.1), kw=dict(bias=-2.5, Vpi=5.0, loss_dB=3, ER_dB=26)),
    "mzm_1pol_scalar_y": dict(func="MZM", gv=GV_N, inp=("qpsk", 74, (1024,), "complex128", 0.03), el=2.0, kw=dict(pol="y", Vpi=4.0)),
    "mzm_2pol_y_bw": dict(func="MZM", gv=GV_N, inp=("qpsk", 75, (2, 1024), "complex128", 0.03), el=(76, 1.5),
                          kw=dict(bias=1.0, loss_dB=2, ER_dB=40, pol="y", BW=40e9)),
    # float64 twin loop (reference devices.py:2425-2486), 1 polarisation only
    "twin_f64_1pol": dict(func="TWIN", gv=GV_A, inp=("noise", 30, (4096,), "complex128", 0.03),
                          kw=dict(length=10, h=1.0, **FIB)),
    "twin_f64_qpsk_8k": dict(func="TWIN", gv=GV_B, inp=("qpsk", 31, (8192,), "complex128", 0.0316),
                             kw=dict(length=20, h=0.5, **SMF)),
}


def case_dt(case) -> float:
    g = case["gv"]
    return 1.0 / (g["R"] * g["sps"])


def case_input(case):
    if "inp" not in case:
        return None, None
    sig = make_input(*case["inp"][:4], amp=case["inp"][4])
    noi = None
    if case.get("noise"):
        noi = make_input(*case["noise"][:4], amp=case["noise"][4])
    return sig, noi


def case_drive(case):
    """Drive voltage of an MZM case: ``(signal, noise | None)``; a scalar drive is returned as it is."""
    el = case["el"]
    if not isinstance(el, tuple):
        return el, None
    n = case["inp"][2][-1]
    v = np.random.default_rng(el[0]).standard_normal(n) * el[1]
    vn = None
    if case.get("el_noise"):
        vn = np.random.default_rng(case["el_noise"][0]).standard_normal(n) * case["el_noise"][1]
    return v, vn
