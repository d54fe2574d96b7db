"""Hostile parameter values through the HIP path and the oracle: print every row whose bits differ.
GPU box only (tests/oracle_bind.py is test infrastructure).

  (default)              every input of every distribution set to each WEIRD value in turn around a sane point,
                         then HOSTILE_COMBOS (300) random mixtures of weird and sane values (seed HOSTILE_SEED)
  HOSTILE_WIDE=1         valid parameters drawn over far wider spans than the synthetic tables
  HOSTILE_ORACLE_ONLY=1  CPU pre-flight: only check that the oracle ends on every row (run this first -- the
                         kernels follow the oracle's decisions, so a row the oracle cannot finish would hang a launch)
"""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_bind  # noqa: E402
from rimphony_amd import api  # noqa: E402

nan, inf = float("nan"), float("inf")
WEIRD = [0.0, -0.0, -1.0, -2.5, -3.0, nan, inf, -inf, 1e-320, 1e-200, 1e-30, 1e-3, 0.5, 1.0, 2.0, 50.0, 1e6, 1e30, 1e200]
SANE = {0: [2.5, 1.0, 1e12, 1e10], 1: [3.0], 2: [2.5, 1.0, 1.0, 1e12, 1e10], 3: [3.0, 5.0, 1.0, 1e10]}


def main():
    L = oracle_bind.load("det")
    ctx = None if os.environ.get("HOSTILE_ORACLE_ONLY") else api.Context(0)     # CPU pre-flight: does the oracle end?
    total_bad = 0
    for kind in (0, 1, 2, 3):
        rows = []
        for j in range(len(SANE[kind])):
            for v in WEIRD:
                r = list(SANE[kind]); r[j] = v
                rows.append((10.0, 0.8, r))
        for sv in WEIRD:
            rows.append((sv, 0.8, list(SANE[kind])))
        for tv in WEIRD + [np.pi / 2, np.pi / 2 + 0.3, 3.0, np.pi]:
            rows.append((10.0, tv, list(SANE[kind])))
        if kind == 3:      # pairs of bad kappa x width
            for a, b in itertools.product([-2.5, -1.0, 1e-200, 1e200, inf], repeat=2):
                rows.append((10.0, 0.8, [a, b, 1.0, 1e10]))
        if os.environ.get("HOSTILE_WIDE"):
            # valid but wide: every parameter over a far larger span than the synthetic tables use
            rng = np.random.default_rng(int(os.environ.get("HOSTILE_SEED", "1000")) + 10 + kind)
            lu = lambda lo, hi: float(np.exp(rng.uniform(np.log(lo), np.log(hi))))
            rows = []
            for _ in range(int(os.environ.get("HOSTILE_COMBOS", "300"))):
                gmin = lu(1, 1e3)
                if kind == 0:
                    r = [rng.uniform(1.01, 8), gmin, lu(gmin * 10, 1e12), lu(10, 1e12)]
                elif kind == 1:
                    r = [lu(0.01, 1e3)]
                elif kind == 2:
                    r = [rng.uniform(1.01, 8), rng.uniform(0, 10), gmin, lu(gmin * 10, 1e12), lu(10, 1e12)]
                else:
                    r = [rng.uniform(1.1, 10), lu(0.1, 100), rng.uniform(0, 10), lu(10, 1e12)]
                th = rng.uniform(0.01, 1.5607)
                rows.append((lu(0.01, 1e5), th if rng.random() < 0.5 else np.pi - th, [float(v) for v in r]))
        # random combinations: every input drawn from the weird values with probability 1/3, else sane-ish
        rng = np.random.default_rng(int(os.environ.get("HOSTILE_SEED", "1000")) + kind)
        for _ in range(0 if os.environ.get("HOSTILE_WIDE") else int(os.environ.get("HOSTILE_COMBOS", "300"))):
            pick = lambda sane: float(rng.choice(WEIRD)) if rng.random() < 1 / 3 else sane
            r = [pick(v * float(np.exp(rng.normal(0, 0.3)))) for v in SANE[kind]]
            rows.append((pick(float(np.exp(rng.uniform(np.log(0.1), np.log(1e4))))), pick(float(rng.uniform(0.05, 1.52))), r))
        if os.environ.get("HOSTILE_ORACLE_ONLY"):
            import time
            s = np.array([r[0] for r in rows]); th = np.array([r[1] for r in rows])
            params = [np.array([r[2][j] for r in rows]) for j in range(len(SANE[kind]))]
            t0 = time.time()
            ref = oracle_bind.batch(L, kind, s, th, params, 0xFF, nthreads=8)
            print("kind", kind, "rows", len(rows), "oracle %.1f s" % (time.time() - t0), "finite", int(np.isfinite(ref).sum()), flush=True)
            continue
        n = len(rows)
        s = np.array([r[0] for r in rows]); th = np.array([r[1] for r in rows])
        params = [np.array([r[2][j] for r in rows]) for j in range(len(SANE[kind]))]
        got, st = ctx.compute_batch(kind, s, th, params, 0xFF, want_status=True)
        print("kind", kind, "gpu done", flush=True)
        ref = oracle_bind.batch(L, kind, s, th, params, 0xFF, nthreads=16)
        same = (got.view(np.uint64) == ref.view(np.uint64)) | (np.isnan(got) & np.isnan(ref))
        bad = np.flatnonzero(~same.all(axis=1))
        total_bad += len(bad)
        print("kind", kind, "rows", n, "rows with a differing slot:", len(bad), flush=True)
        for i in bad:
            print("   ", rows[i], "slots", np.flatnonzero(~same[i]).tolist(), "gpu", got[i][~same[i]][:3], "ref", ref[i][~same[i]][:3])
    print("TOTAL differing rows", total_bad)


if __name__ == "__main__":
    main()
