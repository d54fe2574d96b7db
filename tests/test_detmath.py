"""Accuracy of the deterministic elementary functions (rimphony_amd/csrc/detmath.h)
against mpmath: they replace libm/ocml on BOTH sides of the parity tests, so they
are pinned here independently (<= 1.5 ulp)."""
import ctypes
import math
import os
import subprocess

import mpmath as mp
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include "detmath.h"
double t_exp(double x){return rim_exp(x);}
double t_log(double x){return rim_log(x);}
double t_log10(double x){return rim_log10(x);}
double t_pow(double x,double y){return rim_pow(x,y);}
double t_lgamma(double x){return rim_lgamma_pos(x);}
double t_sin(double x){double s,c;rim_sincos(x,&s,&c);return s;}
double t_cos(double x){double s,c;rim_sincos(x,&s,&c);return c;}
/* mismatches of rim_div_by against the division operator over n pairs */
long t_div_by_mismatches(const double *a, const double *b, long n)
{ long bad = 0; for (long i = 0; i < n; i++) { if (rim_div_by(a[i], b[i], 1.0 / b[i]) != a[i] / b[i]) bad++; } return bad; }
double t_pow15(double x){return rim_pow15(x);}
double t_log10_region(double x){return rim_log10_region(x);}
double t_sqrt(double x){return rim_sqrt(x);}
double t_pow_pos(double x,double y){return rim_pow_pos(x,y);}
double t_pow_normal(double x,double y){return rim_pow_normal(x,y);}
double t_div_by(double a,double b){return rim_div_by(a,b,1.0/b);}
double t_cbrt(double x){return rim_cbrt_normal(x);}
double t_rgamma(double x){return rim_rgamma_near(x);}
double t_third(double x,double j){double o[4];rim_third_powers(x,o);return o[(int)j];}
double t_rqrt4(double x){return rim_rqrt4_normal(x);}
double t_powexp(double x,double y){return rim_powexp_normal(x,y,-x*1e-3);}
void t_meissel_roots(double y,double *o){rim_meissel_roots(y,o,o+1,o+2);}
'''


@pytest.fixture(scope="module")
def dm(tmp_path_factory):
    d = tmp_path_factory.mktemp("dm")
    c = d / "dm.c"
    c.write_text(SRC)
    so = d / "dm.so"
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-mfma", "-msse4.1", "-fPIC", "-shared",
                    "-I", os.path.join(ROOT, "rimphony_amd", "csrc"), str(c), "-o", str(so), "-lm"], check=True)
    L = ctypes.CDLL(str(so))
    for n in ("t_exp", "t_log", "t_log10", "t_lgamma", "t_sin", "t_cos"):
        getattr(L, n).restype = ctypes.c_double
        getattr(L, n).argtypes = [ctypes.c_double]
    L.t_pow.restype = ctypes.c_double
    L.t_pow.argtypes = [ctypes.c_double] * 2
    return L


def ulp_err(got, exact):
    e = float(exact)
    if e == 0:
        return abs(got)
    return float(abs(mp.mpf(got) - exact) / mp.mpf(math.ulp(e)))


def worst(f, ref, xs):
    return max(ulp_err(f(float(x)), ref(mp.mpf(float(x)))) for x in xs)


def test_exp_log(dm):
    mp.mp.prec = 200
    rng = np.random.default_rng(1)
    assert worst(dm.t_exp, mp.exp, np.concatenate([rng.uniform(-700, 700, 1500), rng.uniform(-1, 1, 500)])) < 1.0
    assert worst(dm.t_log, mp.log, np.concatenate([np.exp(rng.uniform(-700, 700, 1500)), rng.uniform(0.5, 2, 500)])) < 0.6
    assert worst(dm.t_log10, mp.log10, np.concatenate([np.exp(rng.uniform(-700, 700, 1500)), rng.uniform(0.5, 2, 500)])) < 0.6
    assert dm.t_exp(0.0) == 1.0 and dm.t_log(1.0) == 0.0 and dm.t_log10(1000.0) == 3.0
    assert dm.t_exp(800.0) == math.inf and dm.t_exp(-800.0) == 0.0
    assert math.isnan(dm.t_log(-1.0)) and dm.t_log(0.0) == -math.inf


def test_pow(dm):
    mp.mp.prec = 200
    rng = np.random.default_rng(2)
    w = 0.0
    for i in range(3000):
        if i < 1000:
            x, y = float(np.exp(rng.uniform(0, math.log(1e12)))), float(rng.uniform(-5, 0))
        elif i < 2000:
            x, y = float(rng.uniform(0, 1)), float(rng.uniform(0, 3))
        else:
            x, y = float(np.exp(rng.uniform(0, math.log(1e15)))), 1 / 3
        w = max(w, ulp_err(dm.t_pow(x, y), mp.power(mp.mpf(x), mp.mpf(y))))
    assert w < 1.5
    assert dm.t_pow(2, 10) == 1024.0 and dm.t_pow(-2, 3) == -8.0 and math.isnan(dm.t_pow(-2, 0.5))
    assert dm.t_pow(0, 2) == 0.0 and dm.t_pow(0, -1) == math.inf and dm.t_pow(0.0, 0.0) == 1.0


def test_lgamma_sincos(dm):
    mp.mp.prec = 200
    rng = np.random.default_rng(3)
    xs = np.concatenate([np.exp(rng.uniform(math.log(30), math.log(1e15), 1500)), rng.uniform(16, 100, 500)])
    assert worst(dm.t_lgamma, mp.loggamma, xs) < 0.8
    # small arguments: absolute accuracy is what the 2F1 normalisation needs
    for x in (1.0, 1.5, 2.0, 2.5, 3.0, 0.7, 4.3):
        assert abs(dm.t_lgamma(x) - float(mp.loggamma(x))) < 5e-15
    th = np.concatenate([rng.uniform(-4, 4, 1500), rng.uniform(-1e5, 1e5, 1500), rng.uniform(0, 1.6, 1000)])
    assert worst(dm.t_sin, mp.sin, th) < 1.5
    assert worst(dm.t_cos, mp.cos, th) < 1.5


def test_div_by_is_the_correctly_rounded_quotient(dm):
    """rim_div_by (3 operations, reciprocal supplied) must return exactly a / b for the divisors it is used
    with in the kernels: harmonic numbers (integers and non-integers >= 30) and the literal denominators."""
    rng = np.random.default_rng(5)
    n = 2_000_000
    b = np.concatenate([30. + np.floor(rng.random(n) * 1e6), 30. + rng.random(n) * 1e7,
                        np.full(n // 2, 0.10321920e8), np.full(n // 2, 0.1476034560e10), np.full(n // 2, 40320.)])
    a = np.concatenate([rng.random(n) * b[:n], b[n:2 * n] * (1. - rng.random(n)),
                        np.exp(rng.random(3 * (n // 2)) * 200. - 100.) * (rng.random(3 * (n // 2)) - 0.5)])
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    dm.t_div_by_mismatches.restype = ctypes.c_long
    dm.t_div_by_mismatches.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
    assert dm.t_div_by_mismatches(a.ctypes.data, b.ctypes.data, len(a)) == 0


def test_pow15(dm):
    dm.t_pow15.restype = ctypes.c_double
    dm.t_pow15.argtypes = [ctypes.c_double]
    mp.mp.prec = 200
    rng = np.random.default_rng(6)
    for x in np.exp(rng.random(2000) * 60. - 40.):
        ref = mp.mpf(float(x)) ** mp.mpf(1.5)
        got = dm.t_pow15(float(x))
        assert abs((mp.mpf(got) - ref) / ref) < 3e-16


def test_cbrt(dm):
    """The cube root of the Debye expansion (bessel.c:180 calls pow(x, 1./3.)): below 1 ulp over the whole normal
    range, exact on perfect cubes; the exponent 0.333...31 of the reference's call is itself 1.85e-17 ln x away."""
    dm.t_cbrt.restype = ctypes.c_double
    dm.t_cbrt.argtypes = [ctypes.c_double]
    mp.mp.prec = 200
    rng = np.random.default_rng(9)
    xs = np.concatenate([np.exp(rng.uniform(-700., 700., 3000)), 30. * np.exp(rng.uniform(0., 34.5, 3000)),
                         [2.2250738585072014e-308, 1.7976931348623157e308, 1., 8., 27.]])
    w = worst(dm.t_cbrt, lambda x: mp.cbrt(x), xs)
    assert w < 1.0, w
    for k in (2., 3., 5., 10., 1e5, 31., 12345.):
        assert dm.t_cbrt(k * k * k) == k


def test_rgamma_third_powers_rqrt4_powexp(dm):
    """The other lock-step leaf functions (detmath.h): 1 / Gamma on (-8.5, 9.5) -- a few ulp, exact zeros at the poles,
    1 / (m - 1)! at the integers --, the four third powers, y^(-1/4) with the Meissel roots built on it, and x^y exp(e)."""
    for n in ("t_rgamma", "t_rqrt4"):
        getattr(dm, n).restype = ctypes.c_double
        getattr(dm, n).argtypes = [ctypes.c_double]
    for n in ("t_third", "t_powexp"):
        getattr(dm, n).restype = ctypes.c_double
        getattr(dm, n).argtypes = [ctypes.c_double] * 2
    dm.t_meissel_roots.restype = None
    dm.t_meissel_roots.argtypes = [ctypes.c_double, ctypes.POINTER(ctypes.c_double)]
    mp.mp.prec = 200
    rng = np.random.default_rng(10)
    zs = rng.uniform(-8.4, 9.4, 3000)
    zs = zs[np.abs(zs - np.rint(zs)) > 1e-3]            # relative accuracy next to the zeros is checked separately
    assert worst(dm.t_rgamma, mp.rgamma, zs) < 4.0
    for m in range(-8, 1):
        assert dm.t_rgamma(float(m)) == 0.
        z = m + 1e-9
        assert abs(dm.t_rgamma(z) / float(mp.rgamma(mp.mpf(z))) - 1.) < 1e-14
    for m in range(1, 10):
        assert abs(dm.t_rgamma(float(m)) * math.factorial(m - 1) - 1.) < 5e-16
    hs = np.concatenate([np.exp(rng.uniform(-700., 2., 2000)), [5e-324, 1e-310, 2.2250738585072014e-308]])
    for j, e in enumerate((mp.mpf(2) / 3, -mp.mpf(2) / 3, mp.mpf(1) / 3, -mp.mpf(1) / 3)):
        assert worst(lambda x: dm.t_third(x, float(j)), lambda x: x ** e, hs) < 3.0
    assert [dm.t_third(0., float(j)) for j in range(4)] == [0., math.inf, 0., math.inf]
    assert all(math.isnan(dm.t_third(-1., float(j))) for j in range(4))
    ys = np.exp(rng.uniform(math.log(1e-16), math.log(2.), 3000))
    assert worst(dm.t_rqrt4, lambda y: y ** (-mp.mpf(1) / 4), ys) < 1.0
    o = (ctypes.c_double * 3)()
    wz = wu = wf = 0.
    for y in ys[:1000]:
        dm.t_meissel_roots(float(y), o)
        ym = mp.mpf(float(y))
        wz = max(wz, ulp_err(o[0], mp.sqrt(ym)))
        wu = max(wu, ulp_err(o[1], ym ** (-mp.mpf(3) / 2)))
        wf = max(wf, ulp_err(o[2], ym ** (-mp.mpf(1) / 4)))
    assert wz <= 1.0 and wu < 12.0 and wf < 1.0, (wz, wu, wf)       # 1 / Z^3 = w^6 only scales the small V_n sum
    xs, ps = np.exp(rng.uniform(0., 27., 2000)), rng.uniform(-8., -1., 2000)
    w = max(ulp_err(dm.t_powexp(float(x), float(p)), mp.mpf(float(x)) ** mp.mpf(float(p)) * mp.exp(mp.mpf(float(-x * 1e-3))))
            for x, p in zip(xs, ps) if float(x) ** float(p) * math.exp(-x * 1e-3) > 1e-300)       # e = the rounded -x/1000 the C side passes
    assert w < 1.5, w


def test_log10_region(dm):
    """The plain-double log10 of the Bessel region variable: a few ulp (<= 4) on (0, 1], its whole domain."""
    dm.t_log10_region.restype = ctypes.c_double
    dm.t_log10_region.argtypes = [ctypes.c_double]
    mp.mp.prec = 200
    rng = np.random.default_rng(8)
    xs = np.concatenate([np.exp(rng.uniform(math.log(1e-16), 0., 4000)), 1. - np.exp(rng.uniform(math.log(1e-16), math.log(0.5), 1000)),
                         [1.0, 0.5, 0.70710678118654752, 0.7071067811865476]])
    worst = 0.
    for x in xs:
        ref = mp.log10(mp.mpf(float(x)))
        got = dm.t_log10_region(float(x))
        if ref == 0:
            assert got == 0.
            continue
        ulp = abs(float(np.spacing(float(ref))))
        worst = max(worst, float(abs(mp.mpf(got) - ref)) / ulp)
    assert worst <= 4.0, worst


@pytest.mark.gpu
def test_leaf_functions_same_bits_on_gpu(dm):
    """The parity contract: every leaf function gives the same bits compiled by gcc for x86-64 and by hipcc for
    gfx950 -- including the device's bare sqrt sequence across the whole exponent range and the special values."""
    from rimphony_amd import api
    ctx = api.Context(0)
    rng = np.random.default_rng(21)
    n = 200000

    def cpu(name, *cols):
        f = getattr(dm, name)
        f.restype = ctypes.c_double
        f.argtypes = [ctypes.c_double] * len(cols)
        return np.array([f(*[float(c[i]) for c in cols]) for i in range(len(cols[0]))])

    def same(a, b):
        return ((a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))).all()

    # sqrt: random bit patterns (all exponents, both signs, NaNs), the 2^-767 boundary, specials: exact.
    # Below 2^-1000 (next to and inside the subnormal range) the device's bare sequence lacks the library's
    # input scaling and is only required to be within 64 ulp (detmath.h explains why that range is never reached).
    bits = rng.integers(0, 2 ** 64, n, dtype=np.uint64)
    x = np.concatenate([bits.view(np.float64), np.ldexp(1. + rng.random(2000), rng.integers(-1074, -760, 2000)),
                        [0., -0., np.inf, -np.inf, np.nan, 2. ** -767, np.nextafter(2. ** -767, 0), 4.9e-324, 1.7976931348623157e308]])
    got = ctx.detmath_batch("sqrt", x)
    with np.errstate(invalid="ignore"):
        ref = np.sqrt(x)                                  # IEEE sqrt is unique: numpy's is the reference
    tiny = (x > 0) & (x < 2. ** -1000)
    assert same(got[~tiny], ref[~tiny])
    assert tiny.sum() > 300 and (np.abs(got[tiny] - ref[tiny]) <= 64 * np.spacing(ref[tiny])).all()
    sel = slice(0, 20000)
    xs = np.exp(rng.uniform(-700., 700., 20000))
    assert same(ctx.detmath_batch("log", xs), cpu("t_log", xs))
    assert same(ctx.detmath_batch("log10", xs), cpu("t_log10", xs))
    xe = rng.uniform(-745., 709., 20000)
    assert same(ctx.detmath_batch("exp", xe), cpu("t_exp", xe))
    xr = np.exp(rng.uniform(math.log(1e-16), 0., 20000))
    assert same(ctx.detmath_batch("log10_region", xr), cpu("t_log10_region", xr))
    xb, yb = np.exp(rng.uniform(-30., 30., 20000)), rng.uniform(-8., 8., 20000)
    assert same(ctx.detmath_batch("pow", xb, yb), cpu("t_pow", xb, yb))
    xl = np.exp(rng.uniform(math.log(0.5), math.log(1e8), 20000))
    assert same(ctx.detmath_batch("lgamma", xl), cpu("t_lgamma", xl))
    xa = rng.uniform(-1e6, 1e6, 20000)
    assert same(ctx.detmath_batch("sin", xa), cpu("t_sin", xa)) and same(ctx.detmath_batch("cos", xa), cpu("t_cos", xa))
    a, b = np.exp(rng.uniform(-50., 50., 20000)), 30. + np.floor(rng.random(20000) * 1e6)
    assert same(ctx.detmath_batch("div_by", a, b), a / b)
    xc = np.concatenate([np.exp(rng.uniform(-708., 709., 20000)), 30. * np.exp(rng.uniform(0., 34.5, 20000))])
    assert same(ctx.detmath_batch("cbrt", xc), cpu("t_cbrt", xc))
    zg = np.concatenate([rng.uniform(-8.49, 9.49, 20000), np.arange(-8., 10.), np.arange(-8., 10.) + 1e-12])
    assert same(ctx.detmath_batch("rgamma", zg), cpu("t_rgamma", zg))
    ht = np.concatenate([np.exp(rng.uniform(-745., 3., 20000)), [0., 5e-324, 1e-310, -1., np.nan]])
    jt = np.floor(rng.random(len(ht)) * 4.)
    assert same(ctx.detmath_batch("third_powers", ht, jt), cpu("t_third", ht, jt))
    yq = np.exp(rng.uniform(-708., 709., 20000))
    assert same(ctx.detmath_batch("rqrt4", yq), cpu("t_rqrt4", yq))
    xp, yp = np.exp(rng.uniform(0., 30., 20000)), rng.uniform(-9., 9., 20000)
    assert same(ctx.detmath_batch("powexp", xp, yp), cpu("t_powexp", xp, yp))
    ctx.close()


def test_pow_pos_is_pow_on_its_domain(dm):
    for f in (dm.t_pow_pos, dm.t_pow_normal):
        f.restype = ctypes.c_double
        f.argtypes = [ctypes.c_double] * 2
    rng = np.random.default_rng(9)
    xs = np.concatenate([np.exp(rng.uniform(-700., 700., 20000)), [1., 1., 2., 1e-310, 5e-324]])
    ys = np.concatenate([rng.uniform(-12., 12., 20000), [3.3, 0., 0., 0.5, 0.3333333333333333]])
    for x, y in zip(xs, ys):
        a, b = dm.t_pow_pos(float(x), float(y)), dm.t_pow(float(x), float(y))
        assert a == b or (math.isnan(a) and math.isnan(b)), (x, y, a, b)
        if x >= 2.2250738585072014e-308:          # rim_pow_normal: positive normal bases only
            assert dm.t_pow_normal(float(x), float(y)) == b, (x, y)
