#!/usr/bin/env python3
"""Wall time of the full 8-coefficient path (Symphony x6 + Heyvaerts x2) per distribution."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rimphony_amd import api, workload
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = api.Context(0)
dev = torch.device("cuda", 0)
for cfg in ("cfg2_powerlaw_jI_aI", "cfg3_thermal_8", "cfg4_pitchypl_8", "cfg5_pitchykappa_8"):
    if cfg not in workload.CONFIGS:
        print("no config", cfg, list(workload.CONFIGS)); continue
    kind, mask, s, th, params = workload.make_batch(cfg, n, start=0)
    ds, dth, dp = torch.from_numpy(s).to(dev), torch.from_numpy(th).to(dev), [torch.from_numpy(p).to(dev) for p in params]
    for m, label in ((0x3F, "symphony x6"), (0xC0, "faraday x2"), (0xFF, "all 8")):
        t = time.perf_counter()
        out, _ = ctx.compute_batch_device(kind, ds, dth, dp, m)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        w = ctx.last_work()
        nan = int(torch.isnan(out).sum().item())
        print("%-24s %-12s n %d wall %.3f s  -> %.0f points/s  (symphony samples %.3e, faraday samples %.3e, NaN slots %d)"
              % (cfg, label, n, dt, n / dt, w["samples"], w["faraday_samples"], nan), flush=True)
