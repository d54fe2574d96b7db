#!/usr/bin/env python3
"""Diagnostic: kernel time of consecutive launches (same and different batches)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rimphony_amd import api, workload
P = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ctx = api.Context(0)
dev = torch.device("cuda", 0)
batches = []
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 4
for st in range(NB):
    kind, mask, s, th, params = workload.make_batch("cfg2_powerlaw_jI_aI", P, start=st * P)
    batches.append((torch.from_numpy(s).to(dev), torch.from_numpy(th).to(dev), [torch.from_numpy(p).to(dev) for p in params]))
for rep in range(int(sys.argv[3]) if len(sys.argv) > 3 else 2):
    for i, (s, th, params) in enumerate(batches):
        t = time.perf_counter()
        out, _ = ctx.compute_batch_device(kind, s, th, params, mask)
        ms = ctx.last_symphony_ms()
        w = ctx.last_work()
        print("rep", rep, "batch", i, "kernel ms %.1f" % ms, "samples %.4e" % w["samples"], "passes %.4e" % w["passes"],
              "wall %.3f" % (time.perf_counter() - t), "Gsamples/s %.2f" % (w["samples"] / ms / 1e6), flush=True)
