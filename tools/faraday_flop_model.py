#!/usr/bin/env python3
"""Algorithmic fp64 flops per Faraday (Heyvaerts) integrand sample: the hand count of dev_heyvaerts.h weighted with the
sample mix the ORACLE measures on a prefix of a bench table (CPU only; test infrastructure, not product).

Counting rules (the ones behind the 720 flops of a Symphony sample): FMA = 2; add, mul, div,
sqrt, compare-free selects = 1; elementary functions expanded into their detmath.h operation counts:
  log_dd 52, exp_dd 34, pow = log_dd + 4 + exp_dd = 90, pow_from_log 39, lgamma_stirling 82.
Per sample, by branch (source lines: rimphony_amd/csrc/dev_heyvaerts.h):
  fill_coord_vars (190-199)                          18
  dfdsigma (201-219) = calc_f_derivatives + 2        power law 140, thermal 42, pitchy_pl 250, pitchy_kappa 255
  non-resonant element h_nr / f_nr (254-311)         50
  quasi-resonant, g < 10 (222-252, 279-298):         g and prefactors 38 + one log_dd 52
       + 4 fixed-order series, each pow_from_log 39 + 1 division + 1 product + 10 flops per term
         (rim_div_by 5, term product 1, sum 1, convergence test 3)
  quasi-resonant, g >= 10:                            6 real-order series jobs of which ~4.5 run:
       pow_from_log 39 + gamma_real 130 + 8 flops per term, + sincos 2 x 60 + 40
usage: faraday_flop_model.py [rows] -> one line per 8-coefficient bench config."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_bind
from rimphony_amd import workload

DFDS = {0: 140., 1: 42., 2: 250., 3: 255.}
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 96
L = oracle_bind.load("det")
for cfg, start in (("cfg2_powerlaw_8", 1000000), ("cfg3_thermal_8", 0), ("cfg4_pitchypl_8", 0), ("cfg5_pitchykappa_8", 0)):
    kind, mask, s, th, params = workload.make_batch(cfg, rows, start=start)
    out, c = oracle_bind.batch(L, kind, s, th, params, 0xC0, nthreads=8, want_counters=True)
    n = c["integrand_evals"]
    nr, qi, qj = c["hey_nr_samples"], c["hey_qr_i_samples"], c["hey_qr_jy_samples"]
    # the literal oracle sums 4 (f_qr) or 4 (h_qr: 2 + 2) fixed-order series per I-branch sample and 6 / 4 real-order
    # ones per J/Y sample; the kernel sums each distinct J series once (dev_heyvaerts.h:95-135): ~4.5 per sample
    terms_i = c["hey_series_terms"] / max(c["hey_series_calls"], 1)
    common = 18. + DFDS[kind]
    f_nr = common + 50.
    f_qi = common + 38. + 52. + 4. * (39. + 2. + 10. * terms_i)
    f_qj = common + 38. + 52. + 2 * 60. + 40. + 4.5 * (39. + 130. + 8. * terms_i * 4.)   # J/Y series run ~4x longer (x ~ sigma)
    flops = (nr * f_nr + qi * f_qi + qj * f_qj) / n
    print("%-20s rows %d samples %d  mix nr %.3f qr_i %.3f qr_jy %.3f  terms/series %.2f  flops/sample: nr %.0f qr_i %.0f qr_jy %.0f -> %.0f"
          % (cfg, rows, n, nr / n, qi / n, qj / n, terms_i, f_nr, f_qi, f_qj, flops))
