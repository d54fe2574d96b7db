"""Pins the Heyvaerts (Faraday) part of the oracle: the four 1 % known answers the
reference holds (power_law.rs:209-240, thermal_juettner.rs:174-210) and the
series-based Bessel functions that stand in for the un-vendored special-fun crate."""
import ctypes
import math

import numpy as np
import pytest
import scipy.special as sp

import oracle_bind

CASES = [
    # kind, params, s, theta, stokes, expected
    (0, [2.5, 10., 1e12, 1e10], 1e4, 0.25 * math.pi, 1, 1.89e-9),     # pl_heyvaerts_high_freq_rho_q
    (0, [2.5, 10., 1e12, 1e10], 1e4, 0.25 * math.pi, 2, 5.28e-8),     # pl_heyvaerts_high_freq_rho_v
    (1, [10.], 4e4, 0.4, 1, 4.8081e-11),                               # tj_heyvaerts_high_freq_rho_q
    (1, [0.1], 40., 0.5, 2, 3.064e-4),                                 # tj_heyvaerts_high_freq_rho_v
]


@pytest.mark.parametrize("flavour", ["det", "libm"])
@pytest.mark.parametrize("case", CASES)
def test_faraday_known_answers(flavour, case):
    L = oracle_bind.load(flavour)
    kind, par, s, th, stokes, expected = case
    d, st = oracle_bind.mkdist(L, kind, par)
    assert st == 0
    v = L.rimo_compute_dimensionless(d, 2, stokes, s, th, None)
    assert abs(v - expected) < 0.01 * expected, (v, expected)


def _bind(L):
    for n in ("rimo_bessel_i", "rimo_bessel_jnu", "rimo_bessel_ynu"):
        getattr(L, n).restype = ctypes.c_double
        getattr(L, n).argtypes = [ctypes.c_double, ctypes.c_double]
    L.rimo_gamma_real.restype = ctypes.c_double
    L.rimo_gamma_real.argtypes = [ctypes.c_double]
    return L


def test_series_bessel_functions_against_scipy(oracle):
    L = _bind(oracle)
    for nu in (1 / 3, -1 / 3, 2 / 3, -2 / 3):
        for g in (1e-6, 0.01, 0.3, 1., 3., 9.99):      # the I branch is used for g < 10
            assert abs(L.rimo_bessel_i(nu, g) / sp.iv(nu, g) - 1) < 2e-14
    # The J/Y branch (g >= 10) is only reachable where the quasi-resonant pomega range is limited by
    # sqrt(sigma^2 - sigma0^2) rather than by the physical limit, i.e. sigma < 3, and there g >= 10 needs
    # x = sqrt(sigma^2 - pomega^2 - sigma0^2) <= 0.23 (g = sqrt(8)/3 (sigma - x)^1.5 / sqrt(x); at sigma = 3.1 the
    # physical limit already keeps x >= 0.456, g <= 6).  Orders sigma, sigma - 1 and their negatives: (-3, 3).
    # Checked with margin: orders to +-3.6, x to 1.2.
    for nu in (-0.9, -0.5, -0.2, 0.1, 0.5, 0.99, 1.3, 2.3, 2.7, 3.1, 3.6, -1.4, -2.3, -2.9, -3.6):
        for x in (1e-5, 1e-3, 0.05, 0.25, 0.6, 1.2):
            assert abs(L.rimo_bessel_jnu(nu, x) / sp.jv(nu, x) - 1) < 5e-14, (nu, x)
            if nu > -1:
                # reflection formula: cancellation near half-integers
                assert abs(L.rimo_bessel_ynu(nu, x) / sp.yv(nu, x) - 1) < 1e-10, (nu, x)
    # exactly integer order: documented 2^-26 step off the pole of the reflection formula
    assert abs(L.rimo_bessel_ynu(2.0, 0.1) / sp.yv(2.0, 0.1) - 1) < 1e-6
    for z in (1 / 3, 2 / 3, 4 / 3, 5 / 3, -0.5, -1.5, 2.5, 7.25):
        assert abs(L.rimo_gamma_real(z) / sp.gamma(z) - 1) < 1e-14


def test_fixed_order_bessel_i_against_mpmath(oracle, oracle_libm):
    """I_{+-1/3}, I_{+-2/3}(g) for g < 10 -- the quasi-resonant elements' Bessel functions, `special-fun` in the reference
    (source absent) -- against mpmath, BOTH flavours: what stands in for an absent library must at least be as good as a
    library.  The deterministic flavour (= the kernels: one cube root, tabulated 1 / Gamma, Horner on the correctly
    rounded series coefficients from a degree chosen by q alone, dev_heyvaerts.h rim_iseries4) and the literal flavour
    (pow / Gamma / term recurrence; Gamma from tgammal since round 4 -- with exp(lgamma) it sat 15 to 27 ulp off, one sign
    per order, which the difference I_(-nu) - I_nu amplified into the whole Faraday tail of rounds 2-3) must both stay
    within 9 ulp, 2 ulp rms, and carry no bias beyond 1.5 ulp."""
    import mpmath as mp
    mp.mp.dps = 40
    D, Lm = _bind(oracle), _bind(oracle_libm)
    rng = np.random.default_rng(5)
    gs = np.concatenate([np.exp(rng.uniform(math.log(1e-8), math.log(10.), 1000)), rng.uniform(0., 10., 1000),
                         # both sides of every degree threshold of the Horner evaluation
                         [2 * math.sqrt(q) * (1 + e) for q in (0.02182381681565762, 0.2649524471765791, 1.2326589703399826,
                                                               3.602682493649838, 8.079176729641972, 15.339587183182314)
                          for e in (-1e-12, 0., 1e-12)]])
    for nu in (2 / 3, -2 / 3, 1 / 3, -1 / 3):
        for L in (D, Lm):
            err = []
            for g in gs:
                exact = mp.besseli(mp.mpf(nu), mp.mpf(float(g)))
                err.append(float((mp.mpf(L.rimo_bessel_i(nu, float(g))) - exact) / mp.mpf(math.ulp(float(exact)))))
            err = np.array(err)
            assert np.abs(err).max() < 9.0 and err.std() < 2.0 and abs(err.mean()) < 1.5, (nu, np.abs(err).max(), err.std(), err.mean())


def test_joint_jy_evaluation_against_mpmath(oracle, oracle_libm):
    """J_sigma, Y_sigma, J_(sigma-1), Y_(sigma-1) of the J/Y branch (g >= 10: sigma < ~3.05, x <~ 0.3; checked with margin
    to sigma 3.6, x 1.2) against mpmath, both flavours.  The deterministic flavour (= the kernels) evaluates the four series
    jointly (dev_heyvaerts.h bessel_jy_fast: one power, rim_rgamma_quad, numerator / denominator recurrences); the
    literal one by four separate pow / Gamma / term recurrences.  J to 2e-14; Y carries the cancellation of the reflection
    formula near half-integer orders (as Cephes' yv does), so it is held to 1e-9 of max(|Y|, |J_-nu| / |sin|) (worst case, both flavours alike:
    1.5e-10 at sigma = 1.0000001)."""
    import mpmath as mp
    mp.mp.dps = 40
    rng = np.random.default_rng(9)
    sig = np.concatenate([rng.uniform(0.05, 3.6, 400), [0.5, 1.5, 2.5, 0.3333333333333333, 2.9999999, 1.0000001]])
    xs = np.exp(rng.uniform(math.log(1e-5), math.log(1.2), len(sig)))
    out = (ctypes.c_double * 4)()
    for L in (oracle, oracle_libm):
        L.rimo_bessel_jy_set.restype = None
        L.rimo_bessel_jy_set.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_double * 4]
        worst_j = worst_y = 0.
        for s_, x_ in zip(sig, xs):
            L.rimo_bessel_jy_set(float(s_), float(x_), out)
            for k, nu in ((0, s_), (2, s_ - 1.)):
                ej = mp.besselj(mp.mpf(float(nu)), mp.mpf(float(x_)))
                ey = mp.bessely(mp.mpf(float(nu)), mp.mpf(float(x_)))
                worst_j = max(worst_j, float(abs(mp.mpf(out[k]) - ej) / abs(ej)))
                scale = max(abs(ey), abs(mp.besselj(-mp.mpf(float(nu)), mp.mpf(float(x_))) / mp.sin(mp.pi * mp.mpf(float(nu)))))
                worst_y = max(worst_y, float(abs(mp.mpf(out[k + 1]) - ey) / scale))
        assert worst_j < 2e-14 and worst_y < 1e-9, (worst_j, worst_y)
    # integer orders keep the generic evaluation (the documented 2^-26 step off the pole of the reflection formula)
    oracle.rimo_bessel_jy_set(2.0, 0.1, out)
    assert abs(out[1] / sp.yv(2.0, 0.1) - 1) < 1e-6 and abs(out[0] / sp.jv(2.0, 0.1) - 1) < 1e-14


def test_faraday_via_dispatch(oracle):
    d, _ = oracle_bind.mkdist(oracle, 0, [2.5, 10., 1e12, 1e10])
    out = np.zeros(8)
    oracle.rimo_compute_all_dimensionless.restype = None
    oracle.rimo_compute_all_dimensionless.argtypes = [ctypes.POINTER(oracle_bind.Dist), ctypes.c_double, ctypes.c_double,
                                                      ctypes.POINTER(ctypes.c_double), ctypes.c_void_p]
    oracle.rimo_compute_all_dimensionless(d, 1e4, 0.25 * math.pi, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), None)
    assert np.isfinite(out).all()
    assert abs(out[6] / 1.89e-9 - 1) < 0.01 and abs(out[7] / 5.28e-8 - 1) < 0.01
    # pitchy_pl(k=0) == power_law for the Faraday coefficients too (pitchy_pl.rs:194-201)
    d2, _ = oracle_bind.mkdist(oracle, 2, [2.5, 0., 10., 1e12, 1e10])
    for stokes in (1, 2):
        a = oracle.rimo_compute_dimensionless(d, 2, stokes, 1e2, 0.81, None)
        b = oracle.rimo_compute_dimensionless(d2, 2, stokes, 1e2, 0.81, None)
        assert abs(a / b - 1) < 1e-6
