#!/usr/bin/env python3
"""A/B of two ENVIRONMENT settings of one build on the SAME GPU box, e.g. the round-2 Symphony kernel (one wave per
(point, coefficient): RIMPHONY_SYM_SOLO=1) against the group kernel (the coefficients of a point in lock-step).  Each
setting runs in its own child process (the library reads its environment when a context is created), alternating order,
best of `reps`; exits non-zero when the tables differ.
usage: ab_env.py "VAR=1" "VAR=0" [config] [rows] [mask] [reps] [start]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
a, b = sys.argv[1], sys.argv[2]
cfg = sys.argv[3] if len(sys.argv) > 3 else "cfg2_powerlaw_8"
rows = sys.argv[4] if len(sys.argv) > 4 else "16384"
mask = sys.argv[5] if len(sys.argv) > 5 else "0x3f"
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 2
start = sys.argv[7] if len(sys.argv) > 7 else "0"
child = r'''
import sys, time, hashlib
sys.path.insert(0, %r)
import torch
from rimphony_amd import api, workload
cfg, rows, mask, start = sys.argv[1], int(sys.argv[2]), int(sys.argv[3], 0), int(sys.argv[4])
ctx = api.Context(0); dev = torch.device("cuda", 0)
kind, _, s, th, params = workload.make_batch(cfg, rows, start=start)
d = [torch.from_numpy(x).to(dev) for x in [s, th] + params]
ctx.compute_batch_device(kind, d[0][:256], d[1][:256], [p[:256] for p in d[2:]], mask)
torch.cuda.synchronize()
best = None
for _ in range(2):
    out, _ = ctx.compute_batch_device(kind, d[0], d[1], d[2:], mask)
    torch.cuda.synchronize()
    ms = (ctx.last_symphony_ms() if mask & 0x3f else 0.) + (ctx.last_faraday_ms() if mask & 0xc0 else 0.)
    best = ms if best is None else min(best, ms)
w = ctx.last_work()
print("%%.2f %%s %%d %%d" %% (best, hashlib.md5(out.cpu().numpy().tobytes()).hexdigest(), w["samples"], w["passes"]))
''' % ROOT
res = {a: [], b: []}
for r in range(reps):
    for setting in ((a, b) if r % 2 == 0 else (b, a)):
        env = dict(os.environ)
        for kv in setting.split(","):
            k, v = kv.split("=")
            env[k] = v
        out = subprocess.run([sys.executable, "-c", child, cfg, rows, mask, start], env=env, capture_output=True, text=True, timeout=900)
        if out.returncode:
            print(out.stderr[-2000:]); sys.exit(1)
        ms, md5, samples, passes = out.stdout.strip().split()[-4:]
        res[setting].append((float(ms), md5, int(samples), int(passes)))
for setting in (a, b):
    r0 = res[setting][0]
    print("%-28s kernel ms %s   samples %d passes %d (%.1f samples/pass)  md5 %s" % (setting, " ".join("%.1f" % m[0] for m in res[setting]), r0[2], r0[3], r0[2] / max(r0[3], 1), r0[1]))
md5s = {m[1] for k in (a, b) for m in res[k]}
print("%s %s rows %s mask %s: ratio B/A (best of %d): %.4f   outputs identical: %s" % (cfg, start, rows, mask, reps, min(m[0] for m in res[b]) / min(m[0] for m in res[a]), len(md5s) == 1), flush=True)
if len(md5s) != 1:
    print("FAIL: the tables differ"); sys.exit(2)
