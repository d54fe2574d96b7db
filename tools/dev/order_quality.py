#!/usr/bin/env python3
"""How good is the expensive-first order of the task queue (rimphony_hip.hip: order_bucket)?  One launch with per-coefficient
work counters; per bucket: rows, mean and max cost; the heaviest rows and the bucket they were put in."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rimphony_amd import api, workload
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2_powerlaw_8"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
start = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
ctx = api.Context(0)
dev = torch.device("cuda", 0)
kind, _, s, th, params = workload.make_batch(cfg, n, start=start)
d = [torch.from_numpy(x).to(dev) for x in [s, th] + params]
out, st, work = ctx.compute_batch_device(kind, d[0], d[1], d[2:], 0xFF, want_status=True, want_work=True)
work = work.cpu().numpy().astype(np.float64)
sn, cs = np.sin(th), np.cos(th)
bucket = np.clip(17 - np.floor(np.log2(s)).astype(int), 1, 31)
if kind in (0, 2):          # power-law families: gamma_min is parameter 1
    gmin = params[1]
    nn = np.floor(s * np.abs(sn) + 30.)
    nos = nn / s
    gp = (nos + np.abs(cs) * np.sqrt(np.maximum(nos * nos - sn * sn, 0.))) / (sn * sn)
    bucket = np.where(gp < gmin, 0, bucket)
for name, cols in (("symphony (six coefficients)", slice(0, 6)), ("faraday (two coefficients)", slice(6, 8))):
    c = work[:, cols].sum(axis=1)
    print("%s rows %d  %s: samples per row mean %.3e  p99 %.3e  max %.3e" % (cfg, n, name, c.mean(), np.percentile(c, 99), c.max()))
    for b in sorted(set(bucket)):
        m = bucket == b
        print("   bucket %2d: rows %6d  mean %.3e  p99 %.3e  max %.3e  share of all samples %.3f" % (b, m.sum(), c[m].mean(), np.percentile(c[m], 99), c[m].max(), c[m].sum() / c.sum()))
    top = np.argsort(-c)[:12]
    print("   heaviest rows: " + "  ".join("%d(b%d, %.1fx mean)" % (i, bucket[i], c[i] / c.mean()) for i in top))
