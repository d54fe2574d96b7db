"""diagnostic_symphony_gamma_integral at awkward harmonic numbers (below the resonance, fractional below 30, 1e15+,
negative, NaN, inf): device seam vs oracle.  GPU box only; test infrastructure.  The oracle ends on all of them
(checked on the CPU first), so the wave QAG does."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_bind
from rimphony_amd import api
nan, inf = float("nan"), float("inf")
ns = np.array([0., 1., 2., 5.5, 8., 29., 29.5, 30., 31., 100.5, 1e4, 1e8, 1e15, 1e16, -1., nan, inf, 1e-320, 7.1, 7.18])
L = oracle_bind.load("det")
ctx = api.Context(0)
same = lambda a, b: (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))
total = 0
for kind, par in ((0, [2.7, 1.0, 1e12, 1e10]), (1, [4.0]), (2, [3.1, 1.4, 1.0, 1e12, 1e10]), (3, [3.3, 6.0, 0.8, 1e10])):
    d, st = oracle_bind.mkdist(L, kind, par)
    for s, th in ((10., 0.8), (1e3, 0.05), (0.3, 1.5)):
        for coeff, stokes, lobe in ((0, 0, 0), (1, 0, 0), (0, 2, 1), (1, 2, 0), (0, 1, 0)):
            got = ctx.gamma_integral_batch(kind, par, coeff, stokes, lobe, s, th, ns)
            ref = np.array([L.rimo_gamma_integral(d, coeff, stokes, lobe, s, th, n) for n in ns])
            bad = np.flatnonzero(~same(got, ref))
            total += len(bad)
            for i in bad:
                print("kind %d s=%g th=%g c=%d st=%d lobe=%d n=%r  gpu %r ref %r" % (kind, s, th, coeff, stokes, lobe, ns[i], got[i], ref[i]))
print("TOTAL differing", total)
