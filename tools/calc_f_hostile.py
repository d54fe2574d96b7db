"""calc_f / calc_f_derivatives on awkward (gamma, cos_xi) pairs: device seam vs oracle; prints differing pairs.
GPU box only; test infrastructure."""
import ctypes, itertools, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_bind
from rimphony_amd import api
nan, inf = float("nan"), float("inf")
gs = [0., -0., 0.5, 1., 1.0000000000000002, 1.5, 10., 1e6, 1e12, 1e13, 1e100, 1e300, inf, -1., -inf, nan, 1e-320]
cs = [-1., 1., 0., -0., 0.5, -0.5, 0.9999999999999999, 1.5, -1.5, nan, inf, 1e-320]
L = oracle_bind.load("det")
ctx = api.Context(0)
pairs = np.array(list(itertools.product(gs, cs)))
g, c = pairs[:, 0].copy(), pairs[:, 1].copy()
same = lambda a, b: (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))
total = 0
for kind, par in ((0, [2.7, 1.0, 1e12, 1e10]), (1, [4.0]), (2, [3.1, 1.4, 1.0, 1e12, 1e10]), (3, [3.3, 6.0, 0.8, 1e10])):
    d, st = oracle_bind.mkdist(L, kind, par)
    f, dg, dc = ctx.calc_f_batch(kind, par, g, c, None)
    rf = np.array([L.rimo_calc_f(d, a, b) for a, b in zip(g, c)])
    rdg, rdc = np.empty(len(g)), np.empty(len(g))
    x, y = ctypes.c_double(), ctypes.c_double()
    for i in range(len(g)):
        L.rimo_calc_f_derivatives(d, g[i], c[i], ctypes.byref(x), ctypes.byref(y))
        rdg[i], rdc[i] = x.value, y.value
    bad = np.flatnonzero(~(same(f, rf) & same(dg, rdg) & same(dc, rdc)))
    total += len(bad)
    print("kind", kind, len(g), "pairs,", len(bad), "differ")
    for i in bad[:40]:
        print("   gamma=%r cos_xi=%r  f %r|%r  dfdg %r|%r  dfdcx %r|%r" % (g[i], c[i], f[i], rf[i], dg[i], rdg[i], dc[i], rdc[i]))
print("TOTAL", total)
