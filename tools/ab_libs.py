#!/usr/bin/env python3
"""A/B of two builds of the library on the SAME GPU box (wall time differs by up to ~10 % between MI355X devices, so
builds must never be compared across gpurun calls).  Each build runs in its own child process (the library path is
read at import); alternating order, best of `reps`.
usage: ab_libs.py LIB_A LIB_B [config] [rows] [mask] [reps]
AB_ALLOW_DIFF=1: the two builds are DIFFERENT computations on purpose (a lock-step reformulation that changes a rounding
in kernels and oracle together): report the ratio, say that the tables differ, exit 0."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
a, b = sys.argv[1], sys.argv[2]
cfg = sys.argv[3] if len(sys.argv) > 3 else "cfg2_powerlaw_8"
rows = sys.argv[4] if len(sys.argv) > 4 else "32768"
mask = sys.argv[5] if len(sys.argv) > 5 else "0xC0"
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
child = r'''
import sys, time
sys.path.insert(0, %r)
import torch
from rimphony_amd import api, workload
cfg, rows, mask = sys.argv[1], int(sys.argv[2]), int(sys.argv[3], 0)
ctx = api.Context(0); dev = torch.device("cuda", 0)
kind, _, s, th, params = workload.make_batch(cfg, rows, start=0)
d = [torch.from_numpy(x).to(dev) for x in [s, th] + params]
ctx.compute_batch_device(kind, d[0][:256], d[1][:256], [p[:256] for p in d[2:]], mask)
torch.cuda.synchronize()
best = None
for _ in range(2):
    out, _ = ctx.compute_batch_device(kind, d[0], d[1], d[2:], mask)
    torch.cuda.synchronize()
    ms = (ctx.last_symphony_ms() if mask & 0x3f else 0.) + (ctx.last_faraday_ms() if mask & 0xc0 else 0.)
    best = ms if best is None else min(best, ms)
import hashlib
print("%%.2f %%s" %% (best, hashlib.md5(out.cpu().numpy().tobytes()).hexdigest()))
''' % ROOT
res = {a: [], b: []}
for r in range(reps):
    for lib in ((a, b) if r % 2 == 0 else (b, a)):
        env = dict(os.environ, RIMPHONY_HIP_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "-c", child, cfg, rows, mask], env=env, capture_output=True, text=True, timeout=600)
        if out.returncode:
            print(out.stderr[-2000:]); sys.exit(1)
        ms, md5 = out.stdout.strip().split()[-2:]
        res[lib].append((float(ms), md5))
for lib in (a, b):
    print("%-40s kernel ms %s   output md5 %s" % (os.path.basename(lib), " ".join("%.1f" % m for m, _ in res[lib]), res[lib][0][1]))
md5s = {m for lib in (a, b) for _, m in res[lib]}
print("ratio B/A (best of %d): %.4f   outputs identical: %s" % (reps, min(m for m, _ in res[b]) / min(m for m, _ in res[a]), len(md5s) == 1))
if len(md5s) != 1 and os.environ.get("AB_ALLOW_DIFF") == "1":
    print("tables differ (md5 %s): allowed, AB_ALLOW_DIFF=1" % " / ".join(sorted(md5s)))
elif len(md5s) != 1:
    # a build that moves a bit is not a faster version of the same computation: this tool must not report it as one
    print("FAIL: the tables of the two builds differ (md5 %s)" % " / ".join(sorted(md5s)))
    sys.exit(2)
