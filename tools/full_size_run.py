#!/usr/bin/env python3
"""One table of BASELINE.json at its FULL size on one GPU, streamed through the device entry in tiles (VERDICT round 3,
item 3b): configs[2] = 1e7 thermal rows, all eight coefficients; or one GPU's share of an 8-GPU configuration (configs[3]:
1e7 pitchy_pl rows over 8 GPUs = rows rank, rank + 8, ... -> 1.25e6 rows).  Inputs of a tile are staged in HBM while the
previous tile computes; per tile the table's size-independent properties are checked (every slot finite or NaN with the
NONFINITE status bit, nothing left NOT_COMPUTED) and a status histogram is accumulated with the library's own
rimphony_status_histogram_device.  Prints the rate (whole run, first and last tile), the NaN fraction per slot, the status
histogram, and a checksum (xor of all finite output bit patterns: two runs of the same table must print the same one).
usage: full_size_run.py CONFIG ROWS [--tile N] [--world W --rank R] [--start S]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from rimphony_amd import api, workload

args = sys.argv[1:]
cfg, rows = args[0], int(float(args[1]))
# (--tile: the end of a launch costs about the same whatever the tile's size, so large tiles amortise it: 1e7 thermal rows
# 299 s in 131072-row tiles, 288 s in 1048576-row ones; the committed profiles name the tile they were run with)
opt = {"--tile": 131072, "--world": 1, "--rank": 0, "--start": 0}
for k in list(opt):
    if k in args:
        opt[k] = int(float(args[args.index(k) + 1]))
tile, world, rank, start0 = opt["--tile"], opt["--world"], opt["--rank"], opt["--start"]
NAMES = ["j_I", "alpha_I", "j_Q", "alpha_Q", "j_V", "alpha_V", "rho_Q", "rho_V"]
ctx = api.Context(0)
dev = torch.device("cuda", 0)
mine = np.arange(rank, rows, world) + start0        # this GPU's rows of the table (interleaved sharding)
n = len(mine)
ntiles = (n + tile - 1) // tile
print("%s: %d rows of the %d-row table (rank %d of %d), %d tiles of %d rows" % (cfg, n, rows, rank, world, ntiles, tile), flush=True)


def stage(t):
    idx = mine[t * tile:(t + 1) * tile]
    # the generator is counter-based: any set of rows can be drawn directly
    kind, mask, s, th, params = workload.make_rows(cfg, idx)
    return kind, mask, [torch.from_numpy(x).to(dev, non_blocking=True) for x in [s, th] + params]


hist = np.zeros((8, 8), dtype=np.int64)
nan_count = np.zeros(8, dtype=np.int64)
checksum = np.uint64(0)
tile_rates = []
nxt = stage(0)
torch.cuda.synchronize()
t_all = time.perf_counter()
sym_ms = far_ms = 0.
for t in range(ntiles):
    kind, mask, d = nxt
    t0 = time.perf_counter()
    out, st = ctx.compute_batch_device(kind, d[0], d[1], d[2:], mask, want_status=True)
    if t + 1 < ntiles:
        nxt = stage(t + 1)              # host generation + H2D of the next tile while this one computes
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tile_rates.append(d[0].numel() / dt)
    sym_ms += ctx.last_symphony_ms(); far_ms += ctx.last_faraday_ms()
    o, s_ = out.cpu().numpy(), st.cpu().numpy()
    nan = np.isnan(o)
    assert (np.isfinite(o) | nan).all()
    assert ((s_ & 16) != 0)[nan].all() and ((s_ & 16) == 0)[~nan].all() and ((s_ & 64) == 0).all(), "status bits do not mark the NaN slots"
    nan_count += nan.sum(axis=0)
    checksum ^= np.bitwise_xor.reduce(o[~nan].view(np.uint64)) if (~nan).any() else np.uint64(0)
    hist += ctx.status_histogram(st)
    if t % 8 == 0 or t == ntiles - 1:
        print("  tile %3d / %d: %.1f k rows/s (running mean %.1f k)" % (t + 1, ntiles, tile_rates[-1] / 1e3, (min((t + 1) * tile, n)) / (time.perf_counter() - t_all) / 1e3), flush=True)
total = time.perf_counter() - t_all
print("%s: %d rows in %.1f s = %.2f k eight-coefficient points/s (kernels: Symphony groups %.1f s, Faraday %.1f s); first tile %.2f k, last tile %.2f k, slowest tile %.2f k"
      % (cfg, n, total, n / total / 1e3, sym_ms / 1e3, far_ms / 1e3, tile_rates[0] / 1e3, tile_rates[-1] / 1e3, min(tile_rates) / 1e3))
print("NaN fraction per slot: " + "  ".join("%s %.4f" % (NAMES[k], nan_count[k] / n) for k in range(8)))
print("rows with at least one status bit, per slot (INNER_FAIL OUTER_FAIL CHUNK_CAP STORE_FULL NONFINITE NORM_FAIL NOT_COMPUTED | clean):")
for k in range(8):
    print("  %-8s %s | %d" % (NAMES[k], " ".join("%8d" % int(hist[k, b]) for b in range(7)), int(hist[k, 7])))
print("checksum (xor of the finite outputs' bit patterns): %016x" % int(checksum))
