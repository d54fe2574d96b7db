#!/usr/bin/env python3
"""Region timers of the symphony kernel (diagnostic build librimphony_prof.so, -DRIM_PROF).

Build:  hipcc <flags of rimphony_amd/_build.py> -DRIM_PROF ... -o rimphony_amd/librimphony_prof.so
Run:    RIMPHONY_HIP_LIB=rimphony_amd/librimphony_prof.so python tools/region_profile.py [npoints]
Prints each region's share of the summed per-wave kernel time (cycle counter, lane 0)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rimphony_amd import api, workload
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ctx = api.Context(0)
dev = torch.device("cuda", 0)
kind, mask, s, th, params = workload.make_batch("cfg2_powerlaw_jI_aI", n, start=0)
ds, dth, dp = torch.from_numpy(s).to(dev), torch.from_numpy(th).to(dev), [torch.from_numpy(p).to(dev) for p in params]
ctx.compute_batch_device(kind, ds, dth, dp, mask)
ctx.debug_counters()                      # reset after the warm-up launch
ctx.compute_batch_device(kind, ds, dth, dp, mask)
w = ctx.last_work()
c = ctx.debug_counters()
names = ["kernel (wave lifetime)", "integrand (f call in wave_qag)", "  bessel pair", "    region select (log10)", "    debye bodies",
         "    meissel bodies", "  distribution term", "request setup (sym_order, gamma_limits)", "    miller (integer orders)",
         "requests (sym_eval_request)", "empty region (timer cost, once per pass)", "wave_gk31 (+rescale_error)", "qag_after_bisect", "qag_pick",
         "unpark (sync + LDS reload + uniformize)", "      meissel: log + exp_val", "      meissel: exp_factor", "      debye: pow(x, 1/3)"]
print("kernel ms %.1f  samples %d passes %d" % (ctx.last_symphony_ms(), w["samples"], w["passes"]))
for k, nm in enumerate(names):
    print("%-45s %6.2f %%   %8.1f cycles/pass" % (nm, 100. * c[k] / c[0], c[k] / max(w["passes"], 1)))

