#!/usr/bin/env python3
"""Execution frequencies of the marked places of the symphony kernel (diagnostic build with -DRIM_PROF -DRIM_PROF_COUNTS).

Build:  tools/build_prof.sh   (librimphony_hits.so)
Run:    RIMPHONY_HIP_LIB=rimphony_amd/librimphony_hits.so python tools/hit_profile.py [npoints] [config]
Prints how often a wave enters each place, per integrand pass."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rimphony_amd import api, workload
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg2_powerlaw_jI_aI"
ctx = api.Context(0)
dev = torch.device("cuda", 0)
kind, mask, s, th, params = workload.make_batch(cfg, n, start=0)
faraday = len(sys.argv) > 3 and sys.argv[3] == "faraday"
mask = 0xc0 if faraday else mask & 0x3f
ds, dth, dp = torch.from_numpy(s).to(dev), torch.from_numpy(th).to(dev), [torch.from_numpy(p).to(dev) for p in params]
ctx.compute_batch_device(kind, ds, dth, dp, mask)
ctx.debug_counters()                      # reset after the warm-up launch
ctx.compute_batch_device(kind, ds, dth, dp, mask)
w = ctx.last_work()
c = ctx.debug_counters()
names = {0: "integrand pass", 1: "  joint first application", 2: "qag_pick argmax (size > 2)", 3: "  tie-break slow path",
         4: "select: log10_region (per order)", 5: "debye pair", 6: "meissel (per order)", 7: "  small-eps branch", 8: "  Z < 1e-3 series",
         9: "  log branch", 10: "exp_factor |e| < 1e-3", 11: "exp_factor early zero", 12: "exp_factor |e| > 690", 13: "exp_factor exp_bounded",
         14: "beta < 0.1", 15: "beta >= 0.1", 16: "miller recurrence", 17: "select: x > n side", 18: "select: blend zone",
         19: "calc_f inside limits", 20: "calc_f_derivatives inside limits", 21: "request setup (pair)", 22: "integral complete",
         24: "emission f_term", 25: "absorption f_term", 30: "LANES active in the Bessel pair", 26: "LANES needing Debye (either order)",
         27: "LANES needing Meissel, order n", 28: "LANES needing Meissel, order n+1", 29: "LANES in the blend zone (order n+1)"}
if faraday:
    names = {0: "integrand pass", 1: "  joint first application", 2: "qag_pick argmax (size > 2)", 22: "integral complete",
             25: "non-resonant element pass", 26: "quasi-resonant element pass", 27: "  I_{+-1/3,2/3} branch executed (bessel_i_g4)",
             28: "  J/Y branch executed (bessel_jy_set)", 29: "    J/Y jobs run (of 6 per execution)", 24: "    gamma_real shift-up iterations",
             30: "    real-order series term iterations", 31: "  fixed-order (table) series iterations (two terms each)"}
    print("faraday kernel ms %.1f  samples %d passes %d  hits[0] %d" % (ctx.last_faraday_ms(), w["faraday_samples"], w["faraday_passes"], c[0]))
else:
    print("kernel ms %.1f  samples %d passes %d  hits[0] %d" % (ctx.last_symphony_ms(), w["samples"], w["passes"], c[0]))
for k in sorted(names):
    print("%-40s %12d   %7.4f per pass" % (names[k], c[k], c[k] / max(c[0], 1)))
