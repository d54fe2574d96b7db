"""diagnostic_symphony_gamma_integrand on awkward (n, gamma) pairs, outside the resonance window the integrator
stays in: device seam vs oracle; prints differing pairs.  GPU box only; test infrastructure."""
import itertools, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_bind
from rimphony_amd import api
nan, inf = float("nan"), float("inf")
ns = [0., 1., 2., 5.5, 29., 29.5, 30., 31., 100., 100.5, 1e4, 1e8, 1e15, 1e16, -1., nan, inf, 1e-320]
gs = [0., 0.5, 1., 1.0000000000000002, 1.01, 1.5, 3., 10., 100., 1e4, 1e8, 1e12, 1e13, 1e300, inf, -1., nan, 1e-320]
L = oracle_bind.load("det")
ctx = api.Context(0)
pairs = np.array(list(itertools.product(ns, gs)))
n, g = pairs[:, 0].copy(), pairs[:, 1].copy()
same = lambda a, b: (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))
total = 0
for kind, par in ((0, [2.7, 1.0, 1e12, 1e10]), (1, [4.0]), (2, [3.1, 1.4, 1.0, 1e12, 1e10]), (3, [3.3, 6.0, 0.8, 1e10])):
    d, st = oracle_bind.mkdist(L, kind, par)
    for s, th in ((10., 0.8), (1e3, 0.05), (0.3, 1.5), (50., 2.4)):
        for coeff, stokes in ((0, 0), (1, 0), (0, 1), (1, 2)):
            got = ctx.gamma_integrand_batch(kind, par, coeff, stokes, s, th, n, g)
            ref = np.array([L.rimo_gamma_integrand(d, coeff, stokes, s, th, a, b) for a, b in zip(n, g)])
            bad = np.flatnonzero(~same(got, ref))
            total += len(bad)
            for i in bad[:6]:
                print("kind %d s=%g th=%g c=%d st=%d  n=%r gamma=%r  gpu %r ref %r" % (kind, s, th, coeff, stokes, n[i], g[i], got[i], ref[i]))
print("pairs per case", len(n), "TOTAL differing", total)
