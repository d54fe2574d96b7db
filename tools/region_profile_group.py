#!/usr/bin/env python3
"""Region timers of the group kernel (diagnostic build librimphony_prof.so, tools/build_prof.sh).

Run:    RIMPHONY_HIP_LIB=rimphony_amd/librimphony_prof.so python tools/region_profile_group.py [config] [npoints] [mask]
Prints each region's share of the summed per-wave kernel time (cycle counter read by lane 0) and its cycles per
executed pass.  Regions nest as the indentation says."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rimphony_amd import api, workload
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2_powerlaw_8"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
mask = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0x3F
ctx = api.Context(0)
dev = torch.device("cuda", 0)
kind, _, s, th, params = workload.make_batch(cfg, n, start=1000000)
ds, dth, dp = torch.from_numpy(s).to(dev), torch.from_numpy(th).to(dev), [torch.from_numpy(p).to(dev) for p in params]
ctx.compute_batch_device(kind, ds, dth, dp, mask)
ctx.debug_counters()                      # reset after the warm-up launch
ctx.compute_batch_device(kind, ds, dth, dp, mask)
w = ctx.last_work()
c = ctx.debug_counters()
names = {0: "kernel (wave lifetime)",
         24: "post (group_turn + sym_post of the members whose turn it is)",
         9: "entries (sym_eval_group)",
         7: "  setup (order records, gamma limits)",
         18: "  picks (every member whose list changed; stash look-up)",
         13: "    lean_pick / spill_pick",
         19: "  pass preparation (who holds the interval)",
         1: "  shared integrand",
         2: "    bessel pair",
         4: "      debye bodies",
         5: "      meissel bodies",
         6: "    distribution terms (both)",
         20: "  member term",
         11: "  wave_gk31 per member",
         12: "  group_book (picked members)",
         21: "  stash filing",
         22: "  booking from the stash",
         23: "  next integral / first-rule decisions",
         26: "consume (sym_consume of the members that posted)",
         10: "empty region (timer cost, once per sample pass)",
         27: "owner phase (until the queue is empty and the own task done)",
         29: "helper phase (until the wave leaves)",
         28: "  entries evaluated for other waves"}
far = not (mask & 0x3F)            # the Faraday pair in lock-step (RIMPHONY_FARADAY_GROUP=1, mask 0xC0)
np_, ns_ = (w["faraday_passes"], w["faraday_samples"]) if far else (w["passes"], w["samples"])
passes = max(np_, 1)
print("%s mask %#x rows %d: kernel ms %.1f  samples %d passes %d" % (cfg, mask, n, ctx.last_faraday_ms() if far else ctx.last_symphony_ms(), ns_, np_))
for k, nm in names.items():
    print("%-62s %6.2f %%   %8.1f cycles/pass" % (nm, 100. * c[k] / c[0], c[k] / passes))
