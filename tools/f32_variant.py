#!/usr/bin/env python3
"""The fp32-integrand variant (RIMPHONY_PRECISION_F32_INTEGRAND, BASELINE.json configs[4]) against the fp64 path on
the same rows: relative-error distribution, NaN-pattern differences, kernel time of both (same GPU, same process).
The precision is refused by the product (RIMPHONY_ENOTSUP, include/rimphony_hip.h); this tool opens the measurement hook
(RIMPHONY_F32_VARIANT=1 before the context is created), which serves the power-law and thermal distributions.
usage: f32_variant.py [config] [rows]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from rimphony_amd import api, workload
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2_powerlaw_8"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
os.environ["RIMPHONY_F32_VARIANT"] = "1"
ctx = api.Context(0)
dev = torch.device("cuda", 0)
kind, mask, s, th, params = workload.make_batch(cfg, rows, start=0)
mask &= 0x3F                                 # the variant covers the six Symphony coefficients
d = [torch.from_numpy(a).to(dev) for a in [s, th] + params]
res = {}
for name, prec in (("fp64", api.PRECISION_F64), ("fp32-integrand", api.PRECISION_F32_INTEGRAND)):
    ctx.compute_batch_device(kind, d[0][:256], d[1][:256], [p[:256] for p in d[2:]], mask, precision=prec)
    best = None
    for _ in range(2):
        out, _ = ctx.compute_batch_device(kind, d[0], d[1], d[2:], mask, precision=prec)
        torch.cuda.synchronize(dev)
        ms = ctx.last_symphony_ms()
        best = ms if best is None else min(best, ms)
    w = ctx.last_work()
    res[name] = (out.cpu().numpy(), best, w["samples"], w["passes"])
    print("%-16s kernel %.1f ms  samples %.4g  passes %.4g  -> %.1f k points/s (six coefficients)"
          % (name, best, w["samples"], w["passes"], rows / best))
cmp = workload.compare_tables(res["fp32-integrand"][0], res["fp64"][0], mask)
print("%s, %d rows: fp32-integrand vs fp64:" % (cfg, rows))
print("  " + ", ".join("%s %s" % (k, ("%.3g" % v) if isinstance(v, float) else v) for k, v in cmp.items()))
print("  speed-up of the Symphony kernel: %.3f" % (res["fp64"][1] / res["fp32-integrand"][1]))
