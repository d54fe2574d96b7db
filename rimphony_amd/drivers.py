#!/usr/bin/env python3
"""Single-point and demo drivers of the reference, on the batched GPU path.

  demo-powerlaw almostuniform1     examples/demo-powerlaw.rs:64-175    64 smoothly varying power-law points, TSV
  all-pitchykappa-cgs NU B NE THETA KAPPA WIDTH K
                                   examples/all-pitchykappa-cgs.rs:96-133   the 8 cgs coefficients of one point
  one-pitchypl-normalized          examples/one-pitchypl-normalized.rs      rho_Q of the point hard-coded there
  one-powerlaw-normalized          examples/one-powerlaw-normalized.rs      alpha_I of the point hard-coded there

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


def rust_e(x):
    """Rust's `{:e}` for an f64: the shortest digits that round-trip, scientific, no padding (0 -> "0e0")."""
    from decimal import Decimal
    x = float(x)
    if math.isnan(x):
        return "NaN"
    if math.isinf(x):
        return "inf" if x > 0 else "-inf"
    sign, digits, exp = Decimal(repr(x)).as_tuple()
    digits = list(digits)
    while len(digits) > 1 and digits[-1] == 0:
        digits.pop()
        exp += 1
    if digits == [0]:
        return ("-" if sign else "") + "0e0"
    mant = str(digits[0]) + ("." + "".join(map(str, digits[1:])) if len(digits) > 1 else "")
    return "%s%se%d" % ("-" if sign else "", mant, exp + len(digits) - 1)


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


# examples/one-pitchypl-normalized.rs:14-19 and one-powerlaw-normalized.rs:14-17: the points their author hard-coded
ONE_PITCHYPL_NORMALIZED = dict(s=8.0973407678629616e0, theta=7.2687065355210786e-2, p=2.7273434060193211e0,
                               k=2.7016346500930695e0)         # (Faraday, Q)
ONE_POWERLAW_NORMALIZED = dict(s=1.0360583634e3, theta=7.4017422303e-1, p=2.3306843452e0)       # (Absorption, I)


def one_pitchypl_normalized_lines(calc_factory):
    """one-pitchypl-normalized.rs:25-30: rho_Q, dimensionless, printed with {:.18e}."""
    from . import api
    c = ONE_PITCHYPL_NORMALIZED
    calc = calc_factory(api.PitchyPowerLawDistribution(c["p"], c["k"]).gamma_limits(1., 1e12, 1e10))
    return [rust_e18(calc.compute_dimensionless(api.Coefficient.Faraday, api.Stokes.Q, c["s"], c["theta"]))]


def one_powerlaw_normalized_lines(calc_factory):
    """one-powerlaw-normalized.rs:20-44: alpha_I in cgs at nu = 1 GHz with B chosen so that nu / nu_c = S, and the
    same with the units removed ("Inner"); the Symphony value the example compares with is hard-coded to 0."""
    import math
    from . import api
    c = ONE_POWERLAW_NORMALIZED
    nu, n_e = 1e9, 1.
    b = api.TWO_PI * api.MASS_ELECTRON * api.SPEED_LIGHT * nu / (api.ELECTRON_CHARGE * c["s"])
    calc = calc_factory(api.PowerLawDistribution(c["p"]).gamma_limits(1., 1e12, 1e10))
    val = calc.compute_cgs(api.Coefficient.Absorption, api.Stokes.I, nu, b, n_e, c["theta"])
    remove_units = (-2. * api.MASS_ELECTRON * api.SPEED_LIGHT * nu * abs(math.cos(c["theta"]))
                    / (api.TWO_PI * api.ELECTRON_CHARGE) ** 2)
    return ["Inner Symphony: %s   Us: %s" % (rust_e(0. * remove_units), rust_e(val * remove_units)),
            "Outer Symphony: %s   Us: %s" % (rust_e(0.), rust_e(val))]


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    d = sub.add_parser("demo-powerlaw")
    d.add_argument("demoname", choices=["almostuniform1"])
    a = sub.add_parser("all-pitchykappa-cgs")
    for name in ("nu", "b", "n_e", "theta", "kappa", "width", "k"):
        a.add_argument(name, type=float)
    sub.add_parser("one-pitchypl-normalized")
    sub.add_parser("one-powerlaw-normalized")
    args = ap.parse_args(argv)
    from . import api
    ctx = api.Context(0)
    if args.cmd == "demo-powerlaw":
        lines = demo_powerlaw_lines(lambda kind, s, th, params: ctx.compute_batch(kind, s, th, params, api.SLOTS_ALL))
    elif args.cmd == "one-pitchypl-normalized":
        lines = one_pitchypl_normalized_lines(lambda dist: dist.full_calculation(ctx))
    elif args.cmd == "one-powerlaw-normalized":
        lines = one_powerlaw_normalized_lines(lambda dist: dist.full_calculation(ctx))
    else:
        lines = all_pitchykappa_cgs_lines(lambda dist: dist.full_calculation(ctx), args.nu, args.b, args.n_e, args.theta,
                                          args.kappa, args.width, args.k)
    sys.stdout.write("\n".join(lines) + "\n")
    ctx.close()


if __name__ == "__main__":
    main()
