#!/usr/bin/env python3
"""Shape of the Faraday work: passes per inner integral, samples per pass (one launch per table)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from rimphony_amd import api, workload
ctx = api.Context(0)
dev = torch.device("cuda", 0)
for cfg, n in (("cfg2_powerlaw_8", 16384), ("cfg3_thermal_8", 16384), ("cfg4_pitchypl_8", 8192), ("cfg5_pitchykappa_8", 4096)):
    kind, mask, s, th, params = workload.make_batch(cfg, n, start=1000000)
    d = [torch.from_numpy(x).to(dev) for x in [s, th] + params]
    ctx.compute_batch_device(kind, d[0], d[1], d[2:], 0xC0)
    w = ctx.last_work()
    print("%-20s rows %6d  kernel %.1f ms  samples %.3e  passes %.3e  inner integrals %.3e  passes/integral %.2f  samples/pass %.1f  first-rule share %.3f"
          % (cfg, n, ctx.last_faraday_ms(), w["faraday_samples"], w["faraday_passes"], w["faraday_inner_qags"],
             w["faraday_passes"] / max(w["faraday_inner_qags"], 1), w["faraday_samples"] / max(w["faraday_passes"], 1),
             31. * w["faraday_inner_qags"] / max(w["faraday_samples"], 1)))
