#!/usr/bin/env python3
"""Heaviest Faraday task of a batch (diagnostics word 14/15 of rimphony_debug_counters)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rimphony_amd import api, workload
cfg = sys.argv[1]; n = int(sys.argv[2]); start = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ctx = api.Context(0)
dev = torch.device("cuda", 0)
kind, mask, s, th, params = workload.make_batch(cfg, n, start=start)
ds, dth, dp = torch.from_numpy(s).to(dev), torch.from_numpy(th).to(dev), [torch.from_numpy(p).to(dev) for p in params]
t = time.perf_counter()
out, st = ctx.compute_batch_device(kind, ds, dth, dp, 0xC0, want_status=True)
torch.cuda.synchronize()
print("wall %.2f s" % (time.perf_counter() - t), ctx.last_work())
c = ctx.debug_counters()
batches, idx = c[15] >> 24, c[15] & 0xffffff
print("heaviest task: point", idx, "batches", batches, "s", s[idx], "theta", th[idx], "params", [float(p[idx]) for p in params],
      "value", out[idx].cpu().numpy()[6:], "status", st[idx].cpu().numpy()[6:])
print("shared batches", c[8], "helper reqs", c[9], "owner shared reqs", c[10], "owner wait ms", c[11] / 1e5, "eval us/req", c[14] / 100. / max(c[7], 1))
