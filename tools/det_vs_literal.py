#!/usr/bin/env python3
"""The deterministic oracle flavour (= the HIP kernels, bit for bit) against the literal vectors, per slot, next to the
literal flavour's own controls (tests/golden/literal_*.npz: out, out_rev, out_fma) -- CPU only.
usage: det_vs_literal.py [threads] [--faraday-only] [config ...]   ->  profiles/r4_det_vs_literal.txt (stdout)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_bind
from rimphony_amd import workload
argv = [a for a in sys.argv[1:] if a != "--faraday-only"]
far = "--faraday-only" in sys.argv
threads = int(argv[0]) if argv else 8
only = argv[1:] or None
NAMES = ["j_I", "alpha_I", "j_Q", "alpha_Q", "j_V", "alpha_V", "rho_Q", "rho_V"]
L = oracle_bind.load("det")


def line(g, l):
    gn, ln = np.isnan(g), np.isnan(l)
    ok = ~gn & ~ln
    rel = np.abs(g[ok] - l[ok]) / np.abs(l[ok])
    if rel.size == 0:
        return "-"
    return "median %.1e p99 %.1e max %.1e  >1e-6: %3d  NaN here:there %d:%d (both %d)" % (
        np.median(rel), np.percentile(rel, 99), rel.max(), (rel > 1e-6).sum(), (gn & ~ln).sum(), (~gn & ln).sum(), (gn & ln).sum())


for cfg in ("cfg2_powerlaw_jI_aI", "cfg2_powerlaw_8", "cfg3_thermal_8", "cfg4_pitchypl_8", "cfg5_pitchykappa_8"):
    if only and cfg not in only:
        continue
    z = np.load(os.path.join(ROOT, "tests", "golden", "literal_%s.npz" % cfg))
    n, start, mask = int(z["n"]), int(z["start"]), int(z["mask"])
    if far:
        if not mask & 0xC0:
            continue
        mask = 0xC0
    kind, _, s, th, params = workload.make_batch(cfg, n, start=start)
    t0 = time.time()
    got = oracle_bind.batch(L, kind, s, th, params, mask, nthreads=threads)
    print("%s (%d rows, %.0f s)" % (cfg, n, time.time() - t0))
    for k in range(8):
        if not mask & (1 << k):
            continue
        print("  %-8s kernels vs literal : %s" % (NAMES[k], line(got[:, k], z["out"][:, k])))
        for key, label in (("out_fma", "contracted control"), ("out_rev", "reversed-GK control")):
            if key in z.files:
                print("  %-8s %-19s: %s" % ("", label, line(z[key][:, k], z["out"][:, k])))
    sys.stdout.flush()
