#!/usr/bin/env python3
"""Step-by-step GPU probe with progress written to gpurun_out/probe.log
(debugging aid: shows where a run stalls)."""
import faulthandler
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
LOG = open(os.path.join(ROOT, "gpurun_out", "probe.log"), "a", buffering=1)


def log(*a):
    msg = " ".join(str(x) for x in a)
    LOG.write("[%8.2f] %s\n" % (time.time() - T0, msg))
    print(msg, flush=True)


T0 = time.time()
faulthandler.enable(file=LOG)


import numpy as np
import oracle_bind
from rimphony_amd import api, workload

log("imports done")
ctx = api.Context(0)
L = oracle_bind.load("det")
log("context created")
kind, mask, s, th, params = workload.make_batch("cfg2_powerlaw_jI_aI", 4096)

import struct
import threading

HB = ctx.heartbeat(0)
BUSY = {"label": None, "t0": 0.0}


def watcher():
    last = None
    stale = 0
    while True:
        time.sleep(2.0)
        if BUSY["label"] is None:
            last, stale = None, 0
            continue
        w = [int(HB[i]) for i in range(12)]
        dec = lambda u: struct.unpack("<d", struct.pack("<Q", u))[0]
        log("   hb[%s +%.0fs] task+1=%d batches=%d phase=%d passes=%d inner_it=%d chunks=%d n_start=%.6g delta_n=%.6g "
            "lane=%d n=%.17g done=%d stage=%d" % (BUSY["label"], time.time() - BUSY["t0"], w[0], w[1], w[2], w[3], w[4], w[5],
                                          dec(w[6]), dec(w[7]), w[8], dec(w[9]), w[10], w[11]))
        stale = stale + 1 if w == last else 0
        last = w
        if time.time() - BUSY["t0"] > 40:
            log("   giving up on", BUSY["label"], "(stale polls: %d)" % stale)
            LOG.flush()
            os._exit(3)


threading.Thread(target=watcher, daemon=True).start()


def run(idx, m, label):
    idx = np.asarray(idx)
    t = time.time()
    BUSY["label"], BUSY["t0"] = label, t
    out, st = ctx.compute_batch(kind, s[idx], th[idx], [p[idx] for p in params], m, want_status=True)
    BUSY["label"] = None
    dt = time.time() - t
    w = ctx.last_work()
    log(label, "gpu done in %.3fs" % dt, "kernel ms %.2f" % ctx.last_symphony_ms(), "work", w)
    t = time.time()
    ref = oracle_bind.batch(L, kind, s[idx], th[idx], [p[idx] for p in params], m, nthreads=16)
    log(label, "oracle done in %.3fs" % (time.time() - t))
    sel = [k for k in range(8) if m & (1 << k)]
    g, r = out[:, sel], ref[:, sel]
    same = (g.view(np.uint64) == r.view(np.uint64)) | (np.isnan(g) & np.isnan(r))
    with np.errstate(all="ignore"):
        rel = np.where(same, 0, np.abs(g - r) / np.abs(r))
    log(label, "bit-identical %d/%d" % (same.sum(), same.size), "max rel %.3e" % np.nanmax(rel),
        "status nonzero", int((st[:, sel] != 0).sum()))
    if not same.all():
        bad = np.argwhere(~same)[:5]
        for b in bad:
            log("   mismatch point", idx[b[0]], "slot", sel[b[1]], "s", s[idx[b[0]]], "th", th[idx[b[0]]],
                "gpu", repr(g[b[0], b[1]]), "ref", repr(r[b[0], b[1]]), "status", st[b[0], sel[b[1]]])


steps = [([8], 0x01, "1pt jI"), ([8], 0x03, "1pt jI+aI"), ([0], 0x01, "pt0 jI"), (range(8), 0x03, "8pt"),
         (range(96), 0x03, "96pt"), (range(96), 0x3C, "96pt QV"), (range(1024), 0x03, "1024pt"),
         (range(4096), 0x03, "4096pt")]
for idx, m, label in steps:
    log("start", label)
    run(list(idx), m, label)
log("probe finished")
