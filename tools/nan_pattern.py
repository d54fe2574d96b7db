#!/usr/bin/env python3
"""Where the deterministic oracle (= the kernels, bit for bit) and the literal-libm oracle disagree on NaN versus a
number, per slot, on the rows of the committed literal vectors (tests/golden/literal_*.npz) -- and the same with
integrand samples in the subnormal range counted as 0 (an investigation knob of the oracle, off everywhere else).
CPU only (test infrastructure).  usage: nan_pattern.py [config ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_bind
from rimphony_amd import workload

NAMES = ["j_I", "alpha_I", "j_Q", "alpha_Q", "j_V", "alpha_V", "rho_Q", "rho_V"]
cores = min(len(os.sched_getaffinity(0)), 16)
for cfg in sys.argv[1:] or ["cfg3_thermal_8"]:
    z = np.load(os.path.join(ROOT, "tests", "golden", "literal_%s.npz" % cfg))
    kind, _, s, th, params = workload.make_batch(cfg, int(z["n"]), start=int(z["start"]))
    lit = z["out"]
    mask = int(z["mask"])
    res = {}
    for flavour in ("det", "libm"):
        L = oracle_bind.load(flavour)
        for flush in (0, 1):
            L.rimo_set_flush_subnormal_samples(flush)
            res[flavour, flush] = oracle_bind.batch(L, kind, s, th, params, mask, nthreads=cores)
        L.rimo_set_flush_subnormal_samples(0)
    assert np.array_equal(np.isnan(res["libm", 0]), np.isnan(lit))
    print("%s, %d rows: coefficients that are NaN in one flavour and a number in the other" % (cfg, len(s)))
    print("  %-8s | %-29s | %-29s | %s" % ("slot", "as computed: det only / libm only", "subnormal samples = 0: det / libm", "NaN in both (as computed / flushed)"))
    for k in range(8):
        if not mask >> k & 1:
            continue
        row = []
        for flush in (0, 1):
            a, b = np.isnan(res["det", flush][:, k]), np.isnan(res["libm", flush][:, k])
            row.append(((a & ~b).sum(), (~a & b).sum(), (a & b).sum()))
        print("  %-8s | %14d / %-14d | %14d / %-14d | %d / %d" % (NAMES[k], row[0][0], row[0][1], row[1][0], row[1][1], row[0][2], row[1][2]))
    a, b = np.isnan(res["det", 1]), np.isnan(res["libm", 0])
    print("  det flavour WITH the flush against the libm flavour as computed: NaN only in det %d, only in libm %d, both %d"
          % ((a & ~b).sum(), (~a & b).sum(), (a & b).sum()))
    both = np.isfinite(res["det", 1]) & np.isfinite(res["libm", 0])
    with np.errstate(all="ignore"):
        r1 = np.abs(res["det", 1][both] / res["libm", 0][both] - 1.)
        both0 = np.isfinite(res["det", 0]) & np.isfinite(res["libm", 0])
        r0 = np.abs(res["det", 0][both0] / res["libm", 0][both0] - 1.)
    print("  relative difference det vs libm (as computed): median %.2e p99 %.2e max %.2e;  det flushed vs libm: median %.2e p99 %.2e max %.2e"
          % (np.median(r0), np.quantile(r0, 0.99), r0.max(), np.median(r1), np.quantile(r1, 0.99), r1.max()))
    d0, d1 = res["det", 0], res["det", 1]
    both = np.isfinite(d0) & np.isfinite(d1)
    with np.errstate(all="ignore"):
        rel = np.abs(d1[both] / d0[both] - 1.)
    print("  det flavour, flushed vs as computed, where both are numbers: max relative change %.2e" % (np.nanmax(rel) if rel.size else 0.))
