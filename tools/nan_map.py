#!/usr/bin/env python3
"""Where in parameter space the reference's algorithm cannot integrate: NaN fraction per output
slot on a grid of (s, theta) x the distribution's third parameter, from `rows` rows of each eight-coefficient table, with
the status bit that ended the coefficient (every NaN of these tables is a GSL status the reference turns into NaN:
symphony.rs:269, 380; heyvaerts.rs:204-211).  For a crank-out user (examples/crank-out-pitchykappa.rs:184-217): the rows
of a training set that will come out NaN, and why.  Runs on the GPU box.  usage: nan_map.py [rows] > profiles/r3_nan_map.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rimphony_amd import api, workload

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
NAMES = ["j_I", "alpha_I", "j_Q", "alpha_Q", "j_V", "alpha_V", "rho_Q", "rho_V"]
BITS = [(1, "inner quadrature failed"), (2, "outer quadrature failed"), (4, "chunk cap"), (8, "store full"), (32, "norm failed")]
ctx = api.Context(0)
s_edges = np.array([0.1, 1., 10., 100., 1e3, 1e4])
t_edges = np.array([0.05, 0.2, 0.5, 0.9, 1.3, 1.52])
for cfg, third_name, third_idx, third_edges in (("cfg2_powerlaw_8", "p", 0, [1.5, 2.3, 3.2, 4.0]), ("cfg3_thermal_8", "T", 0, [0.1, 1., 10., 100.]),
                                                ("cfg4_pitchypl_8", "k", 1, [0., 1., 2., 3.]), ("cfg5_pitchykappa_8", "kappa", 0, [1.5, 2.5, 3.5, 4.5])):
    kind, mask, s, th, params = workload.make_batch(cfg, rows, start=1000000 if cfg == "cfg2_powerlaw_8" else 0)
    out, status = ctx.compute_batch(kind, s, th, params, mask, want_status=True)
    third = params[third_idx]
    nan = np.isnan(out)
    print("== %s, %d rows: %.1f %% of the rows have at least one NaN coefficient" % (cfg, rows, 100. * nan.any(axis=1).mean()))
    for k in range(8):
        if not nan[:, k].any():
            continue
        why = ", ".join("%s %.2f %%" % (txt, 100. * ((status[:, k] & bit) != 0).mean()) for bit, txt in BITS if ((status[:, k] & bit) != 0).any())
        print("  %-8s NaN in %.2f %% of the rows (%s)%s" % (NAMES[k], 100. * nan[:, k].mean(), why, "" if nan[:, k].mean() >= 1e-3 else "   [grid omitted: < 0.1 %]"))
        if nan[:, k].mean() < 1e-3:
            continue
        for j in range(len(third_edges) - 1):
            sel3 = (third >= third_edges[j]) & (third <= third_edges[j + 1])
            print("    %s in [%g, %g]: NaN %% by theta (rows) x s (columns: %s)" % (third_name, third_edges[j], third_edges[j + 1],
                  " ".join("%g-%g" % (s_edges[i], s_edges[i + 1]) for i in range(5))))
            for a in range(5):
                cells = []
                for b in range(5):
                    sel = sel3 & (th >= t_edges[a]) & (th <= t_edges[a + 1]) & (s >= s_edges[b]) & (s <= s_edges[b + 1])
                    cells.append("%5.1f" % (100. * nan[sel, k].mean()) if sel.sum() else "    -")
                print("      theta %4.2f-%4.2f  %s" % (t_edges[a], t_edges[a + 1], " ".join(cells)))
ctx.close()
