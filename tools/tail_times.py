#!/usr/bin/env python3
"""When does the task queue of a launch run dry, and how long does the launch go on after that?  (-DRIM_TAIL_DIAG build:
tools/build_variant.sh rimphony_amd/librimphony_tail.so -DRIM_TAIL_DIAG; the kernels stamp the 100 MHz wall clock at the
first empty fetch from the queue and at every wave's exit.)
usage: RIMPHONY_HIP_LIB=rimphony_amd/librimphony_tail.so python tools/tail_times.py [config] [rows] [start] [mask]
(mask 0xC0: the Faraday kernel alone -- then the pace of its longest outer quadrature is reported as well)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rimphony_amd import api, workload
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2_powerlaw_8"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
start = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
MASK = int(sys.argv[4], 0) if len(sys.argv) > 4 else 0xFF
ctx = api.Context(0)
dev = torch.device("cuda", 0)
kind, _, s, th, params = workload.make_batch(cfg, n, start=start)
d = [torch.from_numpy(x).to(dev) for x in [s, th] + params]
ctx.compute_batch_device(kind, d[0][:256], d[1][:256], [p[:256] for p in d[2:]], 0xFF)
out, st = ctx.compute_batch_device(kind, d[0], d[1], d[2:], 0xFF, want_status=True)
b = (st.cpu().numpy().astype("int64") >> 16) & 0x7fff
import numpy as np
for name, cols in (("symphony coefficient", slice(0, 6)), ("faraday coefficient", slice(6, 8))):
    x = b[:, cols].ravel()
    print("%s rows %d  batches per %s: median %d  p90 %d  p99 %d  p99.9 %d  max %d;  share of all batches in tasks of >= 64 / 128 / 256 / 512 batches: %.3f %.3f %.3f %.3f;  tasks >= 64 / 128 / 256: %d %d %d"
          % (cfg, n, name, np.median(x), np.percentile(x, 90), np.percentile(x, 99), np.percentile(x, 99.9), x.max(),
             x[x >= 64].sum() / x.sum(), x[x >= 128].sum() / x.sum(), x[x >= 256].sum() / x.sum(), x[x >= 512].sum() / x.sum(),
             (x >= 64).sum(), (x >= 128).sum(), (x >= 256).sum()))
for rep in range(2):
    ctx.compute_batch_device(kind, d[0], d[1], d[2:], MASK)
    torch.cuda.synchronize()
    c = ctx.debug_counters()
    t = ctx.last_tail()
    for name, w0, ms in (("symphony groups", 12, ctx.last_symphony_ms()), ("faraday", 10, ctx.last_faraday_ms())):
        if w0 == 12 and not (MASK & 0x3f):
            continue
        dry, end = c[w0], c[w0 + 1]
        print("%s rows %d  %-16s kernel %.1f ms: the launch goes on for %.1f ms after the queue ran dry (%.1f %% of the kernel)"
              % (cfg, n, name, ms, (end - dry) / 1e5, 100. * ((end - dry) / 1e5) / max(ms, 1e-9)))
    print("   heaviest chains:", {k: v for k, v in t.items() if "heaviest" in k})
    if c[8] and c[9] and MASK == 0xC0:
        rel = lambda w: (int(c[w]) - int(c[10])) / 1e5
        print("   the last Faraday chain of >= 2048 batches to end: began %.1f ms and ended %.1f ms after the queue ran dry (negative: before); "
              "an outer quadrature passed 64 / 512 / 2048 subintervals at %.1f / %.1f / %.1f ms; %d of its batches were published as the champion's"
              % (rel(8), rel(9), rel(12), rel(13), (int(c[2]) - int(c[10])) / 1e5 if c[2] > 10**9 else float("nan"), c[14]))
