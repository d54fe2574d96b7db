#!/usr/bin/env python3
"""MEASURE the fp32 variants against fp64 on one table ("it would cost the instructions it saves" is an
argument, not a number): four legs, each in its own process on a context that owns the device, same rows, six Symphony
coefficients --
  fp64 group     the product's default (group_kernel)
  fp64 solo      RIMPHONY_SYM_SOLO=1: the round-2 kernel the fp32 variants are built on
  fp32 plain     RIMPHONY_F32_VARIANT=1, precision 1, on LIB (the shipped library: v_exp_f32 / v_log_f32 cores, 1e-7 noise);
                 power-law and thermal only
  fp32 smooth    the same on SMOOTH_LIB (tools/build_variant.sh SMOOTH_LIB -DRIM_F32_SMOOTH: fp32 seed + one fp64 correction
                 step for every exp and pow of the integrand); all four distributions
and per fp32 leg the error against fp64 and the NaNs that fp64 does not have.
usage: f32_smooth_measure.py SMOOTH_LIB [config] [rows]"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
smooth = os.path.abspath(sys.argv[1])
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg5_pitchykappa_8"
rows = sys.argv[3] if len(sys.argv) > 3 else "16384"
child = r'''
import sys, os, json
sys.path.insert(0, %r)
import numpy as np, torch
from rimphony_amd import api, workload
cfg, rows, prec, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
ctx = api.Context(0); dev = torch.device("cuda", 0)
kind, mask, s, th, params = workload.make_batch(cfg, rows, start=0)
d = [torch.from_numpy(x).to(dev) for x in [s, th] + params]
ctx.compute_batch_device(kind, d[0][:256], d[1][:256], [p[:256] for p in d[2:]], 0x3f, precision=prec)
best = None
for _ in range(2):
    o, _ = ctx.compute_batch_device(kind, d[0], d[1], d[2:], 0x3f, precision=prec)
    torch.cuda.synchronize()
    ms = ctx.last_symphony_ms()
    best = ms if best is None else min(best, ms)
w = ctx.last_work()
np.save(out, o.cpu().numpy())
print(json.dumps({"ms": best, "samples": w["samples"], "passes": w["passes"]}))
''' % ROOT
legs = [("fp64 group", {}, None, 0), ("fp64 solo", {"RIMPHONY_SYM_SOLO": "1"}, None, 0)]
if cfg in ("cfg2_powerlaw_8", "cfg3_thermal_8"):
    legs.append(("fp32 plain", {"RIMPHONY_F32_VARIANT": "1"}, None, 1))
legs.append(("fp32 smooth", {"RIMPHONY_F32_VARIANT": "1"}, smooth, 1))
import numpy as np
res = {}
for name, env, lib, prec in legs:
    e = dict(os.environ, **env)
    if lib:
        e["RIMPHONY_HIP_LIB"] = lib
    out = "/tmp/f32m_%s.npy" % name.replace(" ", "_")
    r = subprocess.run([sys.executable, "-c", child, cfg, rows, str(prec), out], env=e, capture_output=True, text=True, timeout=900)
    if r.returncode:
        print(name, "FAILED:", r.stderr[-800:]); sys.exit(1)
    res[name] = (json.loads(r.stdout.strip().splitlines()[-1]), np.load(out))
ref_ms = res["fp64 group"][0]["ms"]
f64 = res["fp64 group"][1][:, :6]
assert ((res["fp64 solo"][1][:, :6].view(np.uint64) == f64.view(np.uint64)) | (np.isnan(f64) & np.isnan(res["fp64 solo"][1][:, :6]))).all()
print("%s, %s rows, six Symphony coefficients" % (cfg, rows))
for name, _, _, _ in legs:
    info, o = res[name]
    o = o[:, :6]
    line = "%-12s kernel %8.1f ms = %.2f x fp64 group, %.2f x fp64 solo; samples %.4g passes %.4g" % (
        name, info["ms"], info["ms"] / ref_ms, info["ms"] / res["fp64 solo"][0]["ms"], info["samples"], info["passes"])
    if name.startswith("fp32"):
        both = np.isfinite(o) & np.isfinite(f64)
        rel = np.abs(o[both] - f64[both]) / np.abs(f64[both])
        line += "; vs fp64: median %.1e p99 %.1e max %.1e within 1e-6 %.2f %%; NaN only here %d, only in fp64 %d (of %d)" % (
            np.median(rel), np.percentile(rel, 99), rel.max(), 100. * (rel <= 1e-6).mean(), (np.isnan(o) & ~np.isnan(f64)).sum(),
            (~np.isnan(o) & np.isnan(f64)).sum(), o.size)
    print(line)
