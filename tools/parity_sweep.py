#!/usr/bin/env python3
"""One-off evidence run: GPU vs CPU oracle over larger batches than the test-suite uses (bit equality, NaN
patterns, sample counts).  Needs a GPU and ~3 minutes of 16 host threads.  Test infrastructure, like tests/."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_bind
from rimphony_amd import api, workload
L = oracle_bind.load("det")
ctx = api.Context(0)
cores = min(len(os.sched_getaffinity(0)), 16)
def same(a, b):
    return (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))
for cfg, n, mask in (("cfg2_powerlaw_jI_aI", 8192, 0x03), ("cfg2_powerlaw_8", 768, 0xFF), ("cfg3_thermal_8", 768, 0xFF),
                     ("cfg4_pitchypl_8", 512, 0xFF), ("cfg5_pitchykappa_8", 384, 0xFF)):
    n *= int(os.environ.get("SWEEP_SCALE", "1"))
    kind, _, s, th, params = workload.make_batch(cfg, n, start=int(os.environ.get("SWEEP_START", "300000")))
    t0 = time.perf_counter()
    got, st = ctx.compute_batch(kind, s, th, params, mask, want_status=True)
    tg = time.perf_counter() - t0
    w = ctx.last_work()
    t0 = time.perf_counter()
    ref, ctr = oracle_bind.batch(L, kind, s, th, params, mask, nthreads=cores, want_counters=True)
    tc = time.perf_counter() - t0
    sel = [k for k in range(8) if mask >> k & 1]
    eq = same(got[:, sel], ref[:, sel])
    rel = np.abs(got[:, sel] / ref[:, sel] - 1.)
    rel = rel[np.isfinite(rel)]
    print("%-22s n=%5d slots=%d  identical %d / %d  NaN(gpu) %d NaN(ref) %d  max rel %.1e  gpu %.2fs cpu(%d thr) %.1fs  gpu symphony samples %d"
          % (cfg, n, len(sel), int(eq.sum()), eq.size, int(np.isnan(got[:, sel]).sum()), int(np.isnan(ref[:, sel]).sum()),
             rel.max() if rel.size else 0., tg, cores, tc, w["samples"]), flush=True)
