#!/usr/bin/env python3
"""A/B: cooperative tail on/off (run twice with RIMPHONY_NO_ASSIST unset / =1)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rimphony_amd import api, workload
ctx = api.Context(0)
dev = torch.device("cuda", 0)
def run(start, n, label):
    kind, mask, s, th, params = workload.make_batch("cfg2_powerlaw_jI_aI", n, start=start)
    ds, dth, dp = torch.from_numpy(s).to(dev), torch.from_numpy(th).to(dev), [torch.from_numpy(p).to(dev) for p in params]
    for rep in range(2):
        out, _ = ctx.compute_batch_device(kind, ds, dth, dp, mask)
        ms = ctx.last_symphony_ms()
    c = ctx.debug_counters()
    print(label, "n", n, "kernel ms %.1f" % ms, "pts/s %.0f" % (n / ms * 1e3), "| inner_qags", c[3], "shared_batches", c[8],
          "helper_reqs", c[9], "owner_shared_reqs", c[10], "owner_wait_ms_total %.1f" % (c[11] / 1e5), "polls", c[12],
          "empty_visits", c[13], "eval_us_per_req %.1f" % (c[14] / 100. / max(c[3], 1)), "max_wait_ms %.2f" % (c[15] / 1e5),
          flush=True)
print("NO_ASSIST =", os.environ.get("RIMPHONY_NO_ASSIST"))
run(222883, 1, "outlier")
run(0, 64, "64")
run(0, 1024, "1024")
if os.environ.get("AB_BIG"):
    run(0, 4096, "4096")
    run(0, 16384, "16384")
    run(0, 65536, "65536")
if os.environ.get("AB_OUTLIER_BATCH"):
    run(196608, 65536, "batch3(outlier)")
    run(222883 - 2048, 4096, "4096 around outlier")
if os.environ.get("AB_SCAN"):
    for n in (1, 2, 8, 64, 512):
        run(222883, n, "from outlier")
    for n in (8, 64):
        run(222883 - n + 1, n, "ending at outlier")
