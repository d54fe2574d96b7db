"""ctypes binding of the CPU oracle (oracle/liboracle*.so) for the tests, smoke()
and bench.py's cpu_baseline leg.  Test infrastructure only."""
import ctypes
import os
from ctypes import POINTER, c_double, c_int, c_size_t, c_uint32, c_uint64, c_void_p

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Dist(ctypes.Structure):
    _fields_ = [("kind", c_int), ("par", c_double * 6), ("inv_gamma_cutoff", c_double),
                ("inv_kappa_width", c_double), ("neg_inverse_t", c_double), ("norm", c_double)]


class Counters(ctypes.Structure):
    _fields_ = [(n, c_uint64) for n in (
        "integrand_evals", "gk_evals", "inner_qag_calls", "outer_gk_evals", "outer_qag_calls", "deriv_calls",
        "max_inner_size", "max_outer_size", "bessel_calls", "norm_evals",
        "hey_nr_samples", "hey_qr_i_samples", "hey_qr_jy_samples", "hey_series_terms", "hey_series_calls")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


FN = ctypes.CFUNCTYPE(c_double, c_double, c_void_p)
_cache = {}


def load(flavour="det"):
    # "det" / "libm": the two flavours `make all` builds; any other name is liboracle_<name>.so (`make controls`, `make attr`)
    name = {"det": "liboracle.so", "libm": "liboracle_libm.so"}.get(flavour, "liboracle_%s.so" % flavour)
    if name in _cache:
        return _cache[name]
    path = os.path.join(ROOT, "oracle", name)
    if not os.path.exists(path):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "all"], check=True, stdout=subprocess.DEVNULL)
    L = ctypes.CDLL(path)
    dp = POINTER(c_double)
    L.rimo_bessel_j.restype = c_double; L.rimo_bessel_j.argtypes = [c_double, c_double]
    L.rimo_bessel_dj.restype = c_double; L.rimo_bessel_dj.argtypes = [c_double, c_double]
    L.rimo_bessel_jn_int.restype = c_double; L.rimo_bessel_jn_int.argtypes = [c_int, c_double]
    L.rimo_dist_init.restype = c_int; L.rimo_dist_init.argtypes = [POINTER(Dist), c_int, dp]
    L.rimo_calc_f.restype = c_double; L.rimo_calc_f.argtypes = [POINTER(Dist), c_double, c_double]
    L.rimo_calc_f_derivatives.restype = None
    L.rimo_calc_f_derivatives.argtypes = [POINTER(Dist), c_double, c_double, dp, dp]
    L.rimo_compute_dimensionless.restype = c_double
    L.rimo_compute_dimensionless.argtypes = [POINTER(Dist), c_int, c_int, c_double, c_double, POINTER(Counters)]
    L.rimo_compute_cgs.restype = c_double
    L.rimo_compute_cgs.argtypes = [POINTER(Dist), c_int, c_int, c_double, c_double, c_double, c_double]
    L.rimo_gamma_integrand.restype = c_double
    L.rimo_gamma_integrand.argtypes = [POINTER(Dist), c_int, c_int, c_double, c_double, c_double, c_double]
    L.rimo_gamma_integral.restype = c_double
    L.rimo_gamma_integral.argtypes = [POINTER(Dist), c_int, c_int, c_int, c_double, c_double, c_double]
    L.rimo_symphony_deriv_probe.restype = c_double
    L.rimo_symphony_deriv_probe.argtypes = [POINTER(Dist), c_int, c_int, c_int, c_double, c_double, c_double]
    L.rimo_batch.restype = c_int
    L.rimo_batch.argtypes = [c_int, c_size_t, dp, dp, POINTER(dp), c_uint32, dp, POINTER(Counters), c_int]
    L.rimo_batch_norm.restype = c_int
    L.rimo_batch_norm.argtypes = [c_int, c_size_t, POINTER(dp), dp]
    L.rimo_qag_gk31.restype = c_int
    L.rimo_qag_gk31.argtypes = [FN, c_void_p, c_double, c_double, c_double, c_double, c_size_t, dp, dp,
                                POINTER(c_size_t), POINTER(c_uint64)]
    L.rimo_qag_selftest.restype = c_int
    L.rimo_qag_selftest.argtypes = [c_int, c_double, c_double, c_double, c_double, c_double, c_double, c_size_t,
                                    dp, dp, POINTER(c_size_t)]
    L.rimo_deriv_central.restype = c_int
    L.rimo_deriv_central.argtypes = [FN, c_void_p, c_double, c_double, dp, dp]
    L.rimo_hyperg_2F1_at_1.restype = c_double; L.rimo_hyperg_2F1_at_1.argtypes = [c_double] * 3
    L.rimo_gamma_contribution.restype = c_double
    L.rimo_gamma_contribution.argtypes = [POINTER(Dist), c_int, c_int, c_double, c_double, c_double]
    L.rimo_n_integral.restype = c_int
    L.rimo_n_integral.argtypes = [POINTER(Dist), c_int, c_int, c_int, c_double, c_double, c_double, c_double, dp]
    L.rimo_hey_element.restype = c_double
    L.rimo_hey_element.argtypes = [POINTER(Dist), c_int, c_double, c_double, c_int, c_double, c_double]
    L.rimo_hey_outer_integrand.restype = c_double
    L.rimo_hey_outer_integrand.argtypes = [POINTER(Dist), c_int, c_double, c_double, c_int, c_double]
    L.rimo_highfreq.restype = c_int; L.rimo_highfreq.argtypes = [c_int, dp, c_double, c_double, dp]
    L.rimo_bessel_k012.restype = None; L.rimo_bessel_k012.argtypes = [c_double, dp]
    L.rimo_build_flavour.restype = ctypes.c_char_p
    _cache[name] = L
    return L


def mkdist(L, kind, params):
    d = Dist()
    arr = (c_double * len(params))(*params)
    st = L.rimo_dist_init(ctypes.byref(d), kind, arr)
    return d, st


def _dp(a):
    return a.ctypes.data_as(POINTER(c_double))


def batch(L, kind, s, theta, params, mask=0xFF, nthreads=8, want_counters=False):
    s = np.ascontiguousarray(s, dtype=np.float64)
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    params = [np.ascontiguousarray(p, dtype=np.float64) for p in params]
    n = len(s)
    out = np.zeros((n, 8))
    pp = (POINTER(c_double) * len(params))(*[_dp(p) for p in params])
    c = Counters()
    rc = L.rimo_batch(kind, n, _dp(s), _dp(theta), pp, mask, _dp(out), ctypes.byref(c), nthreads)
    assert rc == 0
    return (out, c.as_dict()) if want_counters else out


def batch_norm(L, kind, params):
    params = [np.ascontiguousarray(p, dtype=np.float64) for p in params]
    n = len(params[0])
    out = np.zeros(n)
    pp = (POINTER(c_double) * len(params))(*[_dp(p) for p in params])
    assert L.rimo_batch_norm(kind, n, pp, _dp(out)) == 0
    return out


def qag(L, f, a, b, epsabs, epsrel, limit=1000):
    r, e, sz, nev = c_double(), c_double(), c_size_t(), c_uint64(0)
    cb = FN(lambda x, _c: f(x))
    st = L.rimo_qag_gk31(cb, None, a, b, epsabs, epsrel, limit, ctypes.byref(r), ctypes.byref(e),
                         ctypes.byref(sz), ctypes.byref(nev))
    return st, r.value, e.value, sz.value, nev.value


def qag_selftest(L, family, p0, p1, a, b, epsabs, epsrel, limit):
    r, e, sz = c_double(), c_double(), c_size_t()
    st = L.rimo_qag_selftest(int(family), p0, p1, a, b, epsabs, epsrel, limit, ctypes.byref(r), ctypes.byref(e),
                             ctypes.byref(sz))
    return st, r.value, e.value, sz.value


def highfreq(L, kind, params, s, theta):
    """{rho_Q, rho_V} of the high-frequency closed forms (power law kind 0 / thermal kind 1)."""
    par = (c_double * len(params))(*[float(p) for p in params])
    out = (c_double * 2)()
    rc = L.rimo_highfreq(kind, par, float(s), float(theta), out)
    assert rc == 0
    return out[0], out[1]


def bessel_k012(L, x):
    k = (c_double * 3)()
    L.rimo_bessel_k012(float(x), k)
    return k[0], k[1], k[2]


def n_integral(L, dist, coeff, stokes, negative_lobe, s, theta, n_lo, n_hi):
    """diagnostic_symphony_n_integral: value, or NaN when the QAG reports an error (the Rust Err)."""
    v = c_double()
    rc = L.rimo_n_integral(ctypes.byref(dist), coeff, stokes, negative_lobe, s, theta, n_lo, n_hi, ctypes.byref(v))
    return v.value if rc == 0 else float("nan")
