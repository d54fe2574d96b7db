#!/usr/bin/env python3
"""The reference's own benchmark cases (benches/powerlaw.rs:28-100): sixteen single-coefficient `compute_cgs` calls
on rows of its Latin square -- all eight coefficients on row 0, then ji_1 jq_2 jv_3 ai_4 aq_1 av_2 fq_3 fv_4.
For each: the value through the HIP path (one 1-point batch per call, as the reference computes them), wall time of
that call, wall time of the same call on one host core through the oracle, and whether the bits agree.
GPU box only; test infrastructure (it loads the oracle)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_bind  # noqa: E402
from rimphony_amd import api  # noqa: E402

NUS = [3e8, 1e9, 3e9, 3e10, 3e11]
N_ES = [1e2, 1e3, 1e4, 1e5, 1e6]
SS = [1e0, 1e1, 1e2, 1e3, 1e4]
THETAS = [0.05, 0.430, 0.810, 1.190, 1.5707]
PS = [1.5, 1.75, 2.5, 3.25, 4.]
LATIN_SQUARE = [1, 4, 2, 3, 0, 3, 1, 0, 4, 2, 0, 3, 1, 2, 4, 2, 0, 4, 1, 3, 4, 2, 3, 0, 1]
E, J, A, F = api.Coefficient.Emission, api.Coefficient.Emission, api.Coefficient.Absorption, api.Coefficient.Faraday
I, Q, V = api.Stokes.I, api.Stokes.Q, api.Stokes.V
CASES = [("ji_0", J, I, 0), ("jq_0", J, Q, 0), ("jv_0", J, V, 0), ("ai_0", A, I, 0), ("aq_0", A, Q, 0), ("av_0", A, V, 0),
         ("fq_0", F, Q, 0), ("fv_0", F, V, 0), ("ji_1", J, I, 1), ("jq_2", J, Q, 2), ("jv_3", J, V, 3), ("ai_4", A, I, 4),
         ("aq_1", A, Q, 1), ("av_2", A, V, 2), ("fq_3", F, Q, 3), ("fv_4", F, V, 4)]


def row(r):
    b = 5 * r
    return (NUS[LATIN_SQUARE[b]], N_ES[LATIN_SQUARE[b + 1]], SS[LATIN_SQUARE[b + 2]], THETAS[LATIN_SQUARE[b + 3]],
            PS[LATIN_SQUARE[b + 4]])


def main():
    ctx = api.Context(0)
    L = oracle_bind.load("det")
    ctx.compute_batch(0, [10.], [0.8], [[2.5], [1.], [1e12], [1e10]], 0xFF)        # load the kernels
    print("%-5s %8s %7s %5s  %-24s %9s %9s  %s" % ("case", "s", "theta", "p", "compute_cgs", "gpu ms", "cpu ms", "bits"))
    for name, coeff, stokes, r in CASES:
        nu, n_e, s, theta, p = row(r)
        bfield = api.TWO_PI * api.MASS_ELECTRON * api.SPEED_LIGHT * nu / (api.ELECTRON_CHARGE * s)
        t0 = time.perf_counter()
        calc = api.PowerLawDistribution(p).gamma_limits(1., 1e12, 1e10).full_calculation(ctx)
        val = calc.compute_cgs(coeff, stokes, nu, bfield, n_e, theta)
        tg = time.perf_counter() - t0
        slot = api.slot_of(coeff, stokes)
        s_eff = nu / (api.ELECTRON_CHARGE * bfield / (api.TWO_PI * api.MASS_ELECTRON * api.SPEED_LIGHT))
        t0 = time.perf_counter()
        ref = oracle_bind.batch(L, 0, [s_eff], [theta], [[p], [1.], [1e12], [1e10]], 1 << slot, nthreads=1)[0, slot]
        tc = time.perf_counter() - t0
        ref_cgs = ref * n_e * nu if coeff == api.Coefficient.Emission else ref * n_e / nu
        same = np.float64(val).view(np.uint64) == np.float64(ref_cgs).view(np.uint64) or (np.isnan(val) and np.isnan(ref_cgs))
        print("%-5s %8.0f %7.4f %5.2f  %-24.16e %9.2f %9.1f  %s" % (name, s, theta, p, val, tg * 1e3, tc * 1e3,
                                                                     "identical" if same else "DIFFER ref %.16e" % ref_cgs), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
