#!/usr/bin/env python3
"""Diagnostic: locate the heaviest point of a range of the bench table by bisection on kernel time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rimphony_amd import api, workload
start, count = int(sys.argv[1]), int(sys.argv[2])
ctx = api.Context(0)
kind, mask, s, th, params = workload.make_batch("cfg2_powerlaw_jI_aI", count, start=start)
def run(lo, hi, m=mask):
    out, st = ctx.compute_batch(kind, s[lo:hi], th[lo:hi], [p[lo:hi] for p in params], m, want_status=True)
    return ctx.last_symphony_ms(), ctx.last_work()["samples"], out, st
lo, hi = 0, count
while hi - lo > 1:
    mid = (lo + hi) // 2
    a = run(lo, mid); b = run(mid, hi)
    print(lo, mid, hi, "ms %.0f %.0f" % (a[0], b[0]), "samples %.3e %.3e" % (a[1], b[1]), flush=True)
    if a[0] > b[0]: hi = mid
    else: lo = mid
i = lo
print("heaviest point: row", start + i, "s", s[i], "theta", th[i], "params", [p[i] for p in params])
for m in (1, 2):
    ms, smp, out, st = run(i, i + 1, m)
    print("slot mask", m, "ms %.0f" % ms, "samples %.3e" % smp, "value", out[0, :2], "status", st[0, :2])
