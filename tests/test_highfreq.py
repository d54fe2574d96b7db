"""High-frequency closed-form Faraday approximations: power_law.rs:133-170,
thermal_juettner.rs:94-142.  The reference pins them with four 1 % known answers (power_law.rs:200-207, 224-231;
thermal_juettner.rs:174-181, 194-201), checked here for the oracle and, on the GPU, for the HIP kernel.  It takes K_nu
from the un-vendored special-fun crate, so the oracle is additionally pinned against (a) scipy.special.kv for the
Bessel functions and (b) an independent numpy transcription of the reference's formulas; the HIP kernel is then
compared bit for bit with the oracle (they share rimphony_amd/csrc/highfreq.h, so that test checks the GPU build, not the formulas)."""
import math

import numpy as np
import pytest
import scipy.special as sp

import oracle_bind

E, ME, C = 4.80320680e-10, 9.1093826e-28, 2.99792458e10


def ref_powerlaw(p, gmin, s, theta):
    """numpy transcription of power_law.rs:150-169"""
    q = (0.0085 * 2. / (p - 2.) * ((s / (math.sin(theta) * gmin ** 2)) ** ((p - 2.) / 2.) - 1.) * (p - 1.)
         / gmin ** (1. - p) * (math.sin(theta) / s) ** ((p + 2.) / 2.))
    v = 0.017 * (math.log(gmin) * (p - 1.)) / ((p + 1.) * gmin ** 2) / s * math.sin(theta)
    return q, v


def ref_thermal(t, s, theta):
    """numpy transcription of thermal_juettner.rs:111-141 with scipy's K_nu"""
    f = 2. * E * E / ME
    it = 1. / t
    k0, k1, k2 = sp.kv(0, it), sp.kv(1, it), sp.kv(2, it)
    q = f * math.sin(theta) ** 2 * (k1 + 6. * t * k2) / (2. * C * s ** 2 * k2)
    v = f * math.cos(theta) * k0 / (C * s * k2)
    return q, v


def test_bessel_k_against_scipy():
    L = oracle_bind.load("det")
    xs = np.concatenate([np.exp(np.linspace(math.log(1e-3), math.log(2.), 300)), np.linspace(2., 60., 300),
                         [1.9999999, 2.0, 2.0000001]])
    worst = 0.
    for x in xs:
        k = oracle_bind.bessel_k012(L, x)
        for nu in range(3):
            ref = sp.kv(nu, x)
            worst = max(worst, abs(k[nu] - ref) / ref)
    assert worst < 5e-14, worst


def test_oracle_matches_reference_formulas():
    L = oracle_bind.load("det")
    rng = np.random.default_rng(11)
    for _ in range(300):
        s = math.exp(rng.uniform(math.log(10.), math.log(1e6)))
        th = rng.uniform(0.05, 1.52)
        p, gmin = rng.uniform(2.1, 4.), math.exp(rng.uniform(math.log(1.5), math.log(30.)))
        q, v = oracle_bind.highfreq(L, 0, [p, gmin, 1e12, 1e10], s, th)
        rq, rv = ref_powerlaw(p, gmin, s, th)
        assert abs(q - rq) <= 1e-11 * abs(rq) and abs(v - rv) <= 1e-12 * abs(rv)
        t = math.exp(rng.uniform(math.log(0.1), math.log(100.)))
        q, v = oracle_bind.highfreq(L, 1, [t], s, th)
        rq, rv = ref_thermal(t, s, th)
        assert abs(q - rq) <= 1e-12 * abs(rq) and abs(v - rv) <= 1e-12 * abs(rv)


# (kind, params, s, theta, slot, expected): the reference's own known-answer tests, tolerance 1 % of EXPECTED
REFERENCE_KNOWN_ANSWERS = [
    (0, [2.5, 10., 1e12, 1e10], 1e4, 0.25 * math.pi, 0, 1.81e-9),      # pl_hs11_high_freq_rho_q
    (0, [2.5, 10., 1e12, 1e10], 1e4, 0.25 * math.pi, 1, 1.19e-8),      # pl_hs11_high_freq_rho_v
    (1, [10.], 4e4, 0.4, 0, 4.8081e-11),                               # tj_approx_high_freq_rho_q
    (1, [0.1], 40., 0.5, 1, 3.064e-4),                                 # tj_approx_high_freq_rho_v
]


def test_reference_known_answers():
    L = oracle_bind.load("det")
    for kind, par, s, th, slot, expected in REFERENCE_KNOWN_ANSWERS:
        got = oracle_bind.highfreq(L, kind, par, s, th)[slot]
        assert abs(got - expected) < 0.01 * expected, (kind, par, slot, got, expected)


@pytest.mark.gpu
def test_gpu_reference_known_answers():
    from rimphony_amd import api
    ctx = api.Context(0)
    for kind, par, s, th, slot, expected in REFERENCE_KNOWN_ANSWERS:
        out = ctx.highfreq_batch(kind, [s], [th], [[v] for v in par])
        assert abs(out[0, slot] - expected) < 0.01 * expected, (kind, par, slot, out[0], expected)
    # through the calculator interface, as the reference's tests call it
    apx = api.PowerLawDistribution(2.5).gamma_limits(10., 1e12, 1e10).high_freq_approximation(ctx)
    q = apx.compute_dimensionless(api.Coefficient.Faraday, api.Stokes.Q, 1e4, 0.25 * math.pi)
    assert abs(q - 1.81e-9) < 0.01 * 1.81e-9
    ctx.close()


def test_huang_shcherbakov_figure6_value():
    """power_law.rs:146-149: n = 2.5, s = 1e4, theta = pi/4, gamma_min = 10 gives 1.8e-9 ("looks about right")."""
    L = oracle_bind.load("det")
    q, _ = oracle_bind.highfreq(L, 0, [2.5, 10., 1e12, 1e10], 1e4, math.pi / 4)
    assert abs(q - 1.8e-9) < 0.05e-9


@pytest.mark.gpu
def test_gpu_highfreq_bit_exact():
    from rimphony_amd import api
    L = oracle_bind.load("det")
    ctx = api.Context(0)
    rng = np.random.default_rng(12)
    n = 4096
    s = np.exp(rng.uniform(math.log(1.), math.log(1e6), n))
    th = rng.uniform(0.05, 1.52, n)
    p, gmin = rng.uniform(1.5, 4., n), np.exp(rng.uniform(0., math.log(30.), n))
    out = ctx.highfreq_batch(api.POWER_LAW, s, th, [p, gmin, np.full(n, 1e12), np.full(n, 1e10)])
    for i in range(0, n, 7):
        q, v = oracle_bind.highfreq(L, 0, [p[i], gmin[i]], s[i], th[i])
        assert np.array_equal(np.array([q, v]).view(np.uint64), out[i].view(np.uint64)), (i, q, v, out[i])
    t = np.exp(rng.uniform(math.log(0.05), math.log(200.), n))
    out = ctx.highfreq_batch(api.THERMAL_JUETTNER, s, th, [t])
    for i in range(0, n, 7):
        q, v = oracle_bind.highfreq(L, 1, [t[i]], s[i], th[i])
        assert np.array_equal(np.array([q, v]).view(np.uint64), out[i].view(np.uint64)), (i, q, v, out[i])
    # hostile values (p = 2 divides by zero, gamma_min <= 0, s <= 0, NaN, inf ...): the same bits as the oracle, NaNs included
    nan, inf = float("nan"), float("inf")
    weird = [0.0, -1.0, 2.0, nan, inf, -inf, 1e-320, 1e-200, 1e200, 1.0]
    for kind, sane in ((api.POWER_LAW, [2.5, 10.]), (api.THERMAL_JUETTNER, [10.])):
        rows = []
        for j in range(len(sane) + 2):
            for w in weird:
                r = [1e4, 0.7] + list(sane)
                r[j] = w
                rows.append(r)
        rows = np.array(rows)
        par = [rows[:, 2 + k].copy() for k in range(len(sane))]
        if kind == api.POWER_LAW:
            par += [np.full(len(rows), 1e12), np.full(len(rows), 1e10)]
        out = ctx.highfreq_batch(kind, rows[:, 0].copy(), rows[:, 1].copy(), par)
        for i, r in enumerate(rows):
            q, v = oracle_bind.highfreq(L, kind, list(r[2:]), r[0], r[1])
            ref = np.array([q, v])
            same = (ref.view(np.uint64) == out[i].view(np.uint64)) | (np.isnan(ref) & np.isnan(out[i]))
            assert same.all(), (kind, list(r), ref, out[i])
    # the calculator interface of the reference: only (Faraday, Q|V) are defined
    calc = api.ThermalJuettnerDistribution(10.).high_freq_approximation(ctx)
    assert math.isnan(calc.compute_dimensionless(api.Coefficient.Emission, api.Stokes.I, 1e5, 0.7))
    assert math.isnan(calc.compute_dimensionless(api.Coefficient.Faraday, api.Stokes.I, 1e5, 0.7))
    q = calc.compute_dimensionless(api.Coefficient.Faraday, api.Stokes.Q, 1e5, 0.7)
    assert q == oracle_bind.highfreq(L, 1, [10.], 1e5, 0.7)[0]
    ctx.close()
