#!/usr/bin/env python3
"""Launch-to-launch spread of the Faraday kernel on the bench rows: kernel time and the part after the queue ran dry
(-DRIM_TAIL_DIAG build), eight launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from rimphony_amd import api, workload
ctx = api.Context(0)
dev = torch.device("cuda", 0)
kind, _, s, th, params = workload.make_batch("cfg2_powerlaw_8", 131072, start=1000000)
d = [torch.from_numpy(x).to(dev) for x in [s, th] + params]
ctx.compute_batch_device(kind, d[0][:256], d[1][:256], [p[:256] for p in d[2:]], 0xFF)
for rep in range(8):
    ctx.compute_batch_device(kind, d[0], d[1], d[2:], 0xFF)
    torch.cuda.synchronize()
    c = ctx.debug_counters()
    print("launch %d: symphony %.1f ms (%.1f after the queue ran dry)   faraday %.1f ms (%.1f after the queue ran dry)"
          % (rep, ctx.last_symphony_ms(), (c[13] - c[12]) / 1e5, ctx.last_faraday_ms(), (c[11] - c[10]) / 1e5), flush=True)
