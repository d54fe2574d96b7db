#!/usr/bin/env python3
"""How many Symphony integrand samples (harmonic n, gamma) the six coefficients of one point have in common: the
reference integrates each coefficient on its own, but j_I, alpha_I, j_Q, alpha_Q visit the same gamma-ranges and start
from the same rule applications, and the Bessel pair -- three quarters of a sample's cost -- depends on (n, gamma) only.
CPU only (oracle investigation knob).  usage: sample_overlap.py [config] [rows]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_bind
from rimphony_amd import workload

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2_powerlaw_8"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 12
L = oracle_bind.load("det")
L.rimo_set_sample_log.restype = None
L.rimo_set_sample_log.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
L.rimo_sample_log_count.restype = ctypes.c_size_t
kind, _, s, th, params = workload.make_batch(cfg, rows, start=0)
CAP = 40_000_000
buf = np.empty(CAP, dtype=np.uint64)
tot = {"all": 0, "union6": 0, "IQ": 0, "unionIQ": 0}
print("%-4s %12s %12s %8s | %12s %12s %8s" % ("row", "samples(6)", "distinct", "ratio", "samples(I,Q)", "distinct", "ratio"))
for r in range(rows):
    d, st = oracle_bind.mkdist(L, kind, [p[r] for p in params])
    keys = {}
    for coeff, stokes in ((0, 0), (1, 0), (0, 1), (1, 1), (0, 2), (1, 2)):
        L.rimo_set_sample_log(buf.ctypes.data, CAP)
        L.rimo_compute_dimensionless(d, coeff, stokes, s[r], th[r], None)
        n = L.rimo_sample_log_count()
        L.rimo_set_sample_log(None, 0)
        if n > CAP:
            keys = None
            break
        keys[coeff, stokes] = np.unique(buf[:n]), n
    if keys is None:
        print("%-4d more than %d samples, skipped" % (r, CAP))
        continue
    n6 = sum(v[1] for v in keys.values())
    u6 = len(np.unique(np.concatenate([v[0] for v in keys.values()])))
    iq = [keys[k] for k in ((0, 0), (1, 0), (0, 1), (1, 1))]
    n4 = sum(v[1] for v in iq)
    u4 = len(np.unique(np.concatenate([v[0] for v in iq])))
    ia = [keys[k] for k in ((0, 0), (1, 0))]
    n2 = sum(v[1] for v in ia)
    u2 = len(np.unique(np.concatenate([v[0] for v in ia])))
    print("%-4d %12d %12d %8.2f | %12d %12d %8.2f | j_I, alpha_I: %d samples, %d distinct (x%.2f)" % (r, n6, u6, n6 / u6, n4, u4, n4 / u4, n2, u2, n2 / u2))
    tot["all"] += n6; tot["union6"] += u6; tot["IQ"] += n4; tot["unionIQ"] += u4
print("total: six coefficients %d samples, %d distinct (x%.2f); j_I, alpha_I, j_Q, alpha_Q %d samples, %d distinct (x%.2f)"
      % (tot["all"], tot["union6"], tot["all"] / max(tot["union6"], 1), tot["IQ"], tot["unionIQ"], tot["IQ"] / max(tot["unionIQ"], 1)))
