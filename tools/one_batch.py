#!/usr/bin/env python3
"""One launch of the hot path over the first N rows of the bench table (profiling target)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rimphony_amd import api, workload
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg2_powerlaw_jI_aI"
sel = int(sys.argv[3], 0) if len(sys.argv) > 3 else None
ctx = api.Context(0)
dev = torch.device("cuda", 0)
kind, mask, s, th, params = workload.make_batch(cfg, n, start=0)
if sel is not None:
    mask = sel
ds, dth, dp = torch.from_numpy(s).to(dev), torch.from_numpy(th).to(dev), [torch.from_numpy(p).to(dev) for p in params]
ctx.compute_batch_device(kind, ds, dth, dp, mask)
w = ctx.last_work()
if mask & 0x3F:
    print("kernel ms %.2f samples %d passes %d inner_qags %d" % (ctx.last_symphony_ms(), w["samples"], w["passes"], w["inner_qags"]))
if mask & 0xC0:
    print("faraday kernel ms %.2f samples %d passes %d inner_qags %d" % (ctx.last_faraday_ms(), w["faraday_samples"], w["faraday_passes"], w["faraday_inner_qags"]))
