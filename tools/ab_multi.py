#!/usr/bin/env python3
"""Several builds / environment settings of the library on the SAME GPU box, round-robin, best of `reps`.
usage: ab_multi.py cfg rows mask reps start SPEC [SPEC ...]     SPEC = path/to/lib.so[,VAR=VALUE...]  ("-" = the in-tree build)
Exits non-zero when the tables differ."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg, rows, mask, reps, start = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
specs = sys.argv[6:]
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
res = {s: [] for s in specs}
for r in range(reps):
    order = specs if r % 2 == 0 else specs[::-1]
    for spec in order:
        parts = spec.split(",")
        env = dict(os.environ)
        if parts[0] != "-":
            env["RIMPHONY_HIP_LIB"] = os.path.abspath(parts[0])
        for kv in parts[1:]:
            k, v = kv.split("=")
            env[k] = v
        out = subprocess.run([sys.executable, "-c", child, cfg, rows, mask, start], env=env, capture_output=True, text=True, timeout=900)
        if out.returncode:
            print(spec, out.stderr[-2000:]); sys.exit(1)
        ms, md5, samples, passes = out.stdout.strip().split()[-4:]
        res[spec].append((float(ms), md5, int(samples), int(passes)))
base = min(m[0] for m in res[specs[0]])
print("%s start %s rows %s mask %s" % (cfg, start, rows, mask))
for spec in specs:
    r0 = res[spec][0]
    print("  %-40s kernel ms %-18s x%.3f  samples/pass %.1f  md5 %s" % (spec, " ".join("%.1f" % m[0] for m in res[spec]), min(m[0] for m in res[spec]) / base, r0[2] / max(r0[3], 1), r0[1][:8]), flush=True)
if len({m[1] for s in specs for m in res[s]}) != 1:
    print("FAIL: the tables differ"); sys.exit(2)
