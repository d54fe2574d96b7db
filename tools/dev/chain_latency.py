#!/usr/bin/env python3
"""One long Faraday task alone on the GPU (cooperative from the start: a launch with fewer tasks than waves starts the
surplus waves as helpers): how long does a batch of its chain take, and where does the time go?  (librimphony_diag.so,
-DRIM_COOP_DIAG counters.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rimphony_amd import api, workload
cfg, row, mask = sys.argv[1], int(sys.argv[2]), int(sys.argv[3], 0)
ctx = api.Context(0)
dev = torch.device("cuda", 0)
kind, _, s, th, params = workload.make_batch(cfg, 1, start=row)
d = [torch.from_numpy(x).to(dev) for x in [s, th] + params]
for rep in range(2):
    out, st = ctx.compute_batch_device(kind, d[0], d[1], d[2:], mask, want_status=True)
    torch.cuda.synchronize()
    ms = ctx.last_faraday_ms() if mask & 0xC0 else ctx.last_symphony_ms()
    c = ctx.debug_counters()
    t = ctx.last_tail()
    w = ctx.last_work()
    nb = max(c[8], 1)
    print("%s row %d mask %#x: kernel %.1f ms; shared batches %d -> %.3f ms per batch; requests by helpers %d, by the owner %d; owner waited %.1f ms in all (%.3f ms per batch); evaluation %.1f us per request (sum %.1f ms); polls %d, empty visits %d; passes %d inner integrals %d"
          % (cfg, row, mask, ms, c[8], ms / nb, c[9], c[10], c[11] / 1e5, c[11] / 1e5 / nb, c[14] / 100. / max(c[9] + c[10], 1), c[14] / 1e5,
             c[12], c[13], w["faraday_passes"], w["faraday_inner_qags"]))
    print("   tail", {k: v for k, v in t.items() if "heaviest" in k}, "out", out.cpu().numpy()[0, 6:], "status", st.cpu().numpy()[0, 6:])
