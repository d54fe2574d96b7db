#!/usr/bin/env python3
"""Single-point and demo drivers of the reference, on the batched GPU path.

  demo-powerlaw almostuniform1     examples/demo-powerlaw.rs:64-175    64 smoothly varying power-law points, TSV
  all-pitchykappa-cgs NU B NE THETA KAPPA WIDTH K
                                   examples/all-pitchykappa-cgs.rs:96-133   the 8 cgs coefficients of one point

Same output text as the Rust programs (`{:.16e}` / `{:.18e}` formatting, same header), so the files feed
neurosynchro's tests unchanged; the 64 demo points go through ONE batched call instead of 64 scalar ones, and
`time_ms(meta)` is the batch time divided by 64.
"""
import argparse
import math
import sys
import time

import numpy as np

from .crank_out import rust_e16

DEMO_HEADER = ("s(lin)\ttheta(lin)\tp(lin)\td(meta)\tpsi(meta)\tn_e(meta)\ttime_ms(meta)\tj_I(res)\talpha_I(res)\t"
               "j_Q(res)\talpha_Q(res)\tj_V(res)\talpha_V(res)\trho_Q(res)\trho_V(res)")
ALL8_LABELS = ("    j_I", "alpha_I", "    j_Q", "alpha_Q", "    j_V", "alpha_V", "  rho_Q", "  rho_V")


def rust_e18(x):
    """Rust's `{:.18e}` for an f64."""
    x = float(x)
    if math.isnan(x):
        return "NaN"
    if math.isinf(x):
        return "inf" if x > 0 else "-inf"
    mant, exp = ("%.18e" % x).split("e")
    return "%se%d" % (mant, int(exp))


def almostuniform1_inputs(steps=64):
    """demo-powerlaw.rs:78-99: gentle linear ramps over a small parameter range."""
    x = np.arange(steps, dtype=np.float64) / float(steps - 1)
    return {"d": x * 3e10, "s": 100. - 10. * x, "theta": 0.5 + 0.1 * x, "psi": 0. + 0.1 * x,
            "n_e": 1e5 - 3e4 * x, "p": 3. - 0.5 * x}


def demo_powerlaw_lines(compute, steps=64):
    """compute(kind, s, theta, params) -> [n, 8]; returns the lines the Rust demo prints."""
    from . import api
    v = almostuniform1_inputs(steps)
    n = steps
    params = [v["p"], np.full(n, 1.), np.full(n, 1e12), np.full(n, 1e10)]      # GAMMA_MIN, GAMMA_MAX, GAMMA_CUTOFF
    t0 = time.perf_counter()
    out = np.asarray(compute(api.POWER_LAW, v["s"], v["theta"], params))
    ms = (time.perf_counter() - t0) * 1e3 / n
    lines = [DEMO_HEADER]
    for i in range(n):
        cols = [v["s"][i], v["theta"][i], v["p"][i], v["d"][i], v["psi"][i], v["n_e"][i], ms] + list(out[i])
        lines.append("\t".join(rust_e16(c) for c in cols))
    return lines


def all_pitchykappa_cgs_lines(calc_factory, nu, b, n_e, theta, kappa, width, k):
    """all-pitchykappa-cgs.rs:96-133 (gamma cutoff 100).  calc_factory(dist) -> calculator with compute_all_cgs."""
    from . import api
    calc = calc_factory(api.PitchyKappaDistribution(kappa, width, k).gamma_cutoff(100.))
    vals = calc.compute_all_cgs(nu, b, n_e, theta)
    return ["%s: %s" % (lab, rust_e18(x)) for lab, x in zip(ALL8_LABELS, vals)]


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    d = sub.add_parser("demo-powerlaw")
    d.add_argument("demoname", choices=["almostuniform1"])
    a = sub.add_parser("all-pitchykappa-cgs")
    for name in ("nu", "b", "n_e", "theta", "kappa", "width", "k"):
        a.add_argument(name, type=float)
    args = ap.parse_args(argv)
    from . import api
    ctx = api.Context(0)
    if args.cmd == "demo-powerlaw":
        lines = demo_powerlaw_lines(lambda kind, s, th, params: ctx.compute_batch(kind, s, th, params, api.SLOTS_ALL))
    else:
        lines = all_pitchykappa_cgs_lines(lambda dist: dist.full_calculation(ctx), args.nu, args.b, args.n_e, args.theta,
                                          args.kappa, args.width, args.k)
    sys.stdout.write("\n".join(lines) + "\n")
    ctx.close()


if __name__ == "__main__":
    main()
