"""All (order, argument) pairs of an awkward grid through pkgw_bessel_j/dj on the device and through the oracle;
prints the pairs whose bits differ.  GPU box only; test infrastructure."""
import itertools, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_bind
from rimphony_amd import api
nan, inf = float("nan"), float("inf")
ns = [0., 1., 2., 5., 29., 30., 31., 12.5, 30.5, 100., 1e3, 1e6, 1e15, 1e16, 1e300, -1., -30., -100.5, nan, inf, -inf, 1e-320, 0.5]
xs = [0., -0., 1e-320, 1e-300, 1e-10, 0.5, 1., 5., 17., 29.9, 30., 31., 100., 1e3, 5e4, 5.1e4, 1e6, 1e15, 1e55, 1e56, 1e300,
      -1., -100., -1e-320, -30., -1e6, nan, inf, -inf]
L = oracle_bind.load("det")
ctx = api.Context(0)
pairs = np.array(list(itertools.product(ns, xs)))
n, x = pairs[:, 0].copy(), pairs[:, 1].copy()
j, dj = ctx.bessel_batch(n, x)
rj = np.array([L.rimo_bessel_j(a, b) for a, b in zip(n, x)])
rdj = np.array([L.rimo_bessel_dj(a, b) for a, b in zip(n, x)])
same = lambda a, b: (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))
bad = np.flatnonzero(~(same(j, rj) & same(dj, rdj)))
print(len(pairs), "pairs,", len(bad), "differ")
for i in bad:
    print("  n=%r x=%r  J gpu %r ref %r   dJ gpu %r ref %r" % (n[i], x[i], j[i], rj[i], dj[i], rdj[i]))
