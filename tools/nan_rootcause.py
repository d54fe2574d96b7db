#!/usr/bin/env python3
"""Why round 2's kernels returned NaN where the reference-like (literal) arithmetic returned a number -- 172 against 11 of
16384 thermal coefficients -- and what closed the gap.

For the Symphony slots of a table this prints, against the committed literal vectors (tests/golden/literal_*.npz):
  * coefficients NaN on one side only, for the deterministic flavour with round 2's rule sums (one product w f per NODE,
    oracle `make controls`: liboracle_pernode.so) and with round 3's (QUADPACK's own pair terms w (f1 + f2): liboracle.so
    = the kernels, bit for bit);
  * for one coefficient that was NaN only in round 2's arithmetic: the quadrature that failed (the fail log of the
    oracle), and its bisection trace in both arithmetics IN UNITS OF THE SUBNORMAL QUANTUM 2^-1074 -- the integrand of
    these gamma-integrals, far out in the harmonic tail, is a bump a few hundred quanta high.
CPU only (test infrastructure).  usage: nan_rootcause.py [config] [threads]"""
import ctypes
import os
import subprocess
import sys
from ctypes import POINTER, c_double, c_int, c_size_t

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_bind
from rimphony_amd import workload

NAMES = ["j_I", "alpha_I", "j_Q", "alpha_Q", "j_V", "alpha_V"]
Q = 4.9406564584124654e-324


class Fail(ctypes.Structure):
    _fields_ = [("n", c_double), ("a", c_double), ("b", c_double), ("result", c_double), ("abserr", c_double),
                ("lobe", c_int), ("status", c_int), ("size", c_int), ("level", c_int)]


def load(path):
    L0 = oracle_bind.load("det")
    L = ctypes.CDLL(path)
    for name in ("rimo_dist_init", "rimo_compute_dimensionless", "rimo_batch", "rimo_gamma_integral"):
        getattr(L, name).restype = getattr(L0, name).restype
        getattr(L, name).argtypes = getattr(L0, name).argtypes
    L.rimo_set_fail_log.argtypes = [POINTER(Fail), c_size_t]; L.rimo_set_fail_log.restype = None
    L.rimo_fail_log_count.restype = c_size_t
    L.rimo_set_qag_trace.argtypes = [POINTER(c_double), c_size_t]; L.rimo_set_qag_trace.restype = None
    L.rimo_qag_trace_rows.restype = c_size_t
    L.rimo_build_flavour.restype = ctypes.c_char_p
    return L


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3_thermal_8"
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "all", "liboracle_pernode.so"], check=True, stdout=subprocess.DEVNULL)
    z = np.load(os.path.join(ROOT, "tests", "golden", "literal_%s.npz" % cfg))
    kind, _, s, th, params = workload.make_batch(cfg, int(z["n"]), start=int(z["start"]))
    lit = z["out"]
    libs = {"per-node products (round 2)": load(os.path.join(ROOT, "oracle", "liboracle_pernode.so")),
            "pair terms (round 3 = kernels)": load(os.path.join(ROOT, "oracle", "liboracle.so"))}
    lit_lib = load(os.path.join(ROOT, "oracle", "liboracle_libm.so"))
    res = {}
    print("%s, %d rows, Symphony slots: coefficients NaN on one side only, deterministic flavour vs literal vectors" % (cfg, len(s)))
    print("  %-32s | %s" % ("rule sums of the det flavour", "  ".join("%-9s" % n for n in NAMES)) + " | total det-only : literal-only")
    for name, L in libs.items():
        out = oracle_bind.batch(L, kind, s, th, params, 0x3F, nthreads=threads)
        res[name] = out
        cells, tot = [], [0, 0]
        for k in range(6):
            a, b = np.isnan(out[:, k]), np.isnan(lit[:, k])
            cells.append("%3d : %-3d" % ((a & ~b).sum(), (~a & b).sum()))
            tot[0] += (a & ~b).sum(); tot[1] += (~a & b).sum()
        print("  %-32s | %s | %d : %d" % (name, "  ".join(cells), tot[0], tot[1]))
    old = res["per-node products (round 2)"]
    rows = np.nonzero(np.isnan(old[:, 4]) & ~np.isnan(lit[:, 4]))[0]
    if not len(rows):
        return
    r = int(rows[0])
    par = [p[r] for p in params]
    print("\nexample: row %d (s = %.6g, theta = %.6g, params %s), j_V: literal %.6g, round-2 arithmetic NaN" % (r, s[r], th[r], par, lit[r, 4]))
    L = libs["per-node products (round 2)"]
    d, _ = oracle_bind.mkdist(L, kind, par)
    buf = (Fail * 8)()
    L.rimo_set_fail_log(buf, 8)
    L.rimo_compute_dimensionless(ctypes.byref(d), 0, 2, s[r], th[r], None)
    cnt = L.rimo_fail_log_count()
    L.rimo_set_fail_log(None, 0)
    for k in range(min(cnt, 2)):
        b = buf[k]
        print("  failed quadrature: %s n = %.17g lobe %d on [%.9g, %.9g]: GSL status %d after %d intervals, result %.4g (%.0f quanta), abserr %.0f quanta"
              % ("gamma-integral" if b.level == 0 else "n-chunk", b.n, b.lobe, b.a, b.b, b.status, b.size, b.result, b.result / Q, b.abserr / Q))
    f0 = buf[0]
    print("  its bisections, in quanta of 2^-1074 (r = result, e = error estimate, asc = resasc of the children):")
    for name, LL in list(libs.items()) + [("literal (glibc libm, qk.c order)", lit_lib)]:
        d, _ = oracle_bind.mkdist(LL, kind, par)
        tb = (c_double * (14 * 200))()
        LL.rimo_set_qag_trace(tb, 200)
        v = LL.rimo_gamma_integral(ctypes.byref(d), 0, 2, f0.lobe, s[r], th[r], f0.n)
        nr = LL.rimo_qag_trace_rows()
        LL.rimo_set_qag_trace(None, 0)
        t = np.array(tb[:14 * nr]).reshape(nr, 14)
        zero_asc = sum(1 for k in range(1, nr) for (e, a) in ((t[k, 5], t[k, 6]), (t[k, 8], t[k, 9])) if a == 0 and e != 0)
        print("   %-32s value %-12.6g %3d bisections; final rt1 %d rt2 %d errsum %.0f tol %.1f; children with resasc = 0 but error != 0: %d"
              % (name, v, nr - 1, t[-1, 10], t[-1, 11], t[-1, 12] / Q, t[-1, 13] / Q, zero_asc))
    print("  (per-node products w f of samples below ~5 quanta round to 0, so resasc -- and with it GSL's rescaling of the error\n"
          "   estimate and its round-off counters -- sees a different integrand than qk.c's w (f1 + f2); sums of whole quanta are\n"
          "   exact in any order, so with the pair terms the rule sums of such intervals carry the reference's bits)")


if __name__ == "__main__":
    main()
