#!/usr/bin/env python3
"""Where do the coefficients of the bench tables end, and what do this library's two result-changing deviations from
the reference cost?  

Part A (GPU): status histogram per slot -- rows whose coefficient ended with each RIMPHONY_ST_* bit -- on the full 1e6-row
table of configs[1] and on 65536 rows of each eight-coefficient table; the rows that hit the chunk cap are written to
gpurun_out/capped_rows.json.
Part B (CPU, oracle = test infrastructure): for capped rows, what the reference's UNBOUNDED marching loop would do --
the oracle re-run with its cap raised from 4096 to 2^18 chunks.
Part C (CPU): the thermal normalisation.  The reference integrates with gsl_integration_qagiu at eps_rel = 1e-5
(thermal_juettner.rs:58-62); here it is the substituted integral at 1e-10 (= T K_2(1/T) to 1e-13).  Every thermal
coefficient is linear in the normalisation, so the relative difference of the norms IS the relative change of all
eight coefficients.  QUADPACK's dqagie -- which GSL's qagiu is a C port of -- is reached through scipy.integrate.quad
with an infinite limit; it stands in for the GSL call.
usage: deviation_report.py gpu | capped FILE | thermal"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from rimphony_amd import workload

BITS = ["INNER_FAIL", "OUTER_FAIL", "CHUNK_CAP", "STORE_FULL", "NONFINITE", "NORM_FAIL", "NOT_COMPUTED", "clean"]
SLOT = ["j_I", "alpha_I", "j_Q", "alpha_Q", "j_V", "alpha_V", "rho_Q", "rho_V"]


def part_gpu():
    import torch
    from rimphony_amd import api
    ctx = api.Context(0)
    dev = torch.device("cuda", 0)
    capped = {}
    for cfg, start, n in (("cfg2_powerlaw_jI_aI", 0, 1000000), ("cfg2_powerlaw_8", 1000000, 65536), ("cfg3_thermal_8", 0, 65536),
                          ("cfg4_pitchypl_8", 0, 65536), ("cfg5_pitchykappa_8", 0, 65536)):
        kind, mask, s, th, params = workload.make_batch(cfg, n, start=start)
        d = [torch.from_numpy(a).to(dev) for a in [s, th] + params]
        t0 = time.time()
        out, st, work = ctx.compute_batch_device(kind, d[0], d[1], d[2:], mask, want_status=True, want_work=True)
        hist = ctx.status_histogram(st)
        torch.cuda.synchronize(dev)
        dt = time.time() - t0
        sth = st.cpu().numpy()
        wk = work.cpu().numpy()
        o = out.cpu().numpy()
        print("\n%s: rows %d..%d, %.1f s" % (cfg, start, start + n, dt))
        print("  %-8s" % "slot" + "".join("%13s" % b for b in BITS) + "%16s%16s" % ("samples median", "samples max"))
        for k in range(8):
            if not mask & (1 << k):
                continue
            print("  %-8s" % SLOT[k] + "".join("%13d" % hist[k, b] for b in range(8)) + "%16d%16d" % (np.median(wk[:, k]), wk[:, k].max()))
        nan_rows = np.isnan(o[:, [k for k in range(8) if mask & (1 << k)]]).any(axis=1).sum()
        print("  rows with any NaN coefficient: %d (%.3f %%)" % (nan_rows, 100. * nan_rows / n))
        rows, slots = np.nonzero((sth & 4) != 0)
        capped[cfg] = {"start": start, "rows": [int(r) for r in rows[:64]], "slots": [int(c) for c in slots[:64]],
                       "count": int(len(rows))}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(capped, open(os.path.join(ROOT, "gpurun_out", "capped_rows.json"), "w"))
    ctx.close()


def part_capped(path):
    import ctypes
    import oracle_bind
    from rimphony_amd import api
    L = oracle_bind.load("det")

    class Tuning(ctypes.Structure):
        _fields_ = [("epsrel_gamma", ctypes.c_double), ("epsrel_n", ctypes.c_double), ("tail_tolerance", ctypes.c_double),
                    ("n_discrete", ctypes.c_int), ("max_chunks", ctypes.c_int), ("hey_max_steps", ctypes.c_int)]
    L.rimo_set_tuning.restype = None
    L.rimo_set_tuning.argtypes = [ctypes.POINTER(Tuning)]
    capped = json.load(open(path))
    for cfg, info in capped.items():
        print("\n%s: %d coefficients hit the chunk cap" % (cfg, info["count"]))
        for r, k in list(zip(info["rows"], info["slots"]))[:6]:
            kind, mask, s, th, params = workload.make_batch(cfg, 1, start=info["start"] + r)
            d, rc = oracle_bind.mkdist(L, kind, [p[0] for p in params])
            co, stk = api.SLOTS[k]
            res = []
            for cap in (4096, 1 << 18):
                L.rimo_set_tuning(ctypes.byref(Tuning(1e-3, 1e-3, 1e5, 30, cap, cap)))
                c = oracle_bind.Counters()
                t0 = time.time()
                v = L.rimo_compute_dimensionless(d, int(co), int(stk), s[0], th[0], ctypes.byref(c))
                res.append((cap, v, c.outer_qag_calls, time.time() - t0))
            L.rimo_set_tuning(None)
            print("  row %d slot %s  s=%.4g theta=%.4g params=%s" % (info["start"] + r, SLOT[k], s[0], th[0], [float("%.4g" % p[0]) for p in params]))
            for cap, v, chunks, dt in res:
                print("      cap %7d chunks: value %-24r outer QAGs %8d  (%.1f s)" % (cap, v, chunks, dt))


def part_thermal():
    from scipy import integrate, special
    rng_T = np.exp(np.linspace(np.log(0.1), np.log(100.), 241))
    rel5, rel10 = [], []
    for T in rng_T:
        exact = T * special.kve(2, 1. / T) * np.exp(-1. / T) if T < 0.2 else T * special.kv(2, 1. / T)
        f = lambda g: g * np.sqrt(g * g - 1.) * np.exp(-g / T)
        v5, _ = integrate.quad(f, 1., np.inf, epsabs=0., epsrel=1e-5, limit=1000)
        rel5.append(v5 / exact - 1.)
    rel5 = np.abs(np.array(rel5))
    print("thermal normalisation integral, QUADPACK dqagie at eps_rel 1e-5 (scipy.integrate.quad; stands in for GSL qagiu,"
          " thermal_juettner.rs:58-62) vs T K_2(1/T), 241 temperatures log-spaced in [0.1, 100]:")
    print("  |relative difference|: median %.2e  p99 %.2e  max %.2e   (requested 1e-5)" % (np.median(rel5), np.quantile(rel5, 0.99), rel5.max()))
    print("  = the relative change of every thermal coefficient if the norm were integrated as the reference does")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "thermal"
    if what == "gpu":
        part_gpu()
    elif what == "capped":
        part_capped(sys.argv[2])
    else:
        part_thermal()
