#!/usr/bin/env python3
"""Region timers of the Faraday kernel (diagnostic build librimphony_prof.so, see tools/region_profile.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rimphony_amd import api, workload
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg2_powerlaw_8"
ctx = api.Context(0)
dev = torch.device("cuda", 0)
kind, mask, s, th, params = workload.make_batch(cfg, n, start=0)
ds, dth, dp = torch.from_numpy(s).to(dev), torch.from_numpy(th).to(dev), [torch.from_numpy(p).to(dev) for p in params]
ctx.compute_batch_device(kind, ds, dth, dp, 0xC0)
ctx.debug_counters()
ctx.compute_batch_device(kind, ds, dth, dp, 0xC0)
w = ctx.last_work()
c = ctx.debug_counters()
print("faraday kernel ms %.1f samples %d passes %d" % (ctx.last_faraday_ms(), w["faraday_samples"], w["faraday_passes"]))
for k, nm in ((0, "kernel (wave lifetime)"), (1, "integrand (f call)"), (18, "  non-resonant passes"), (19, "  quasi-resonant passes"),
              (11, "wave_gk31"), (12, "qag_after_bisect"), (13, "qag_pick"), (14, "unpark")):
    print("%-28s %6.2f %%   %9.1f cycles/pass" % (nm, 100. * c[k] / c[0], c[k] / max(w["faraday_passes"], 1)))
print("quasi-resonant h_qr element calls (passes):", c[22], " with some lane at g >= 10:", c[23], " such lanes:", c[24])
