"""Synthetic parameter batches for the BASELINE.json configurations.

Counter-based so that every consumer (numpy here, any C/HIP driver later) draws
identical bits:  u(i, j) = (splitmix64(seed ^ (16 i + j)) >> 11) * 2^-53  for
point i, parameter j.  Sampling laws follow the reference's Sampler
(test-support/src/lib.rs:31-64: uniform, or log-uniform via exp(U[ln lo, ln hi]))
over the span of its own drivers and golden file.
"""
import numpy as np

SEED_BASE = 20250614
_M64 = (1 << 64) - 1


def splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def uniform01(seed, idx, j):
    with np.errstate(over="ignore"):
        key = np.uint64(seed) ^ (idx.astype(np.uint64) * np.uint64(16) + np.uint64(j))
        r = splitmix64(key)
    return (r >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _lin(u, lo, hi):
    return lo + u * (hi - lo)


def _log(u, lo, hi):
    return np.exp(np.log(lo) + u * (np.log(hi) - np.log(lo)))


# kind ids match include/rimphony_hip.h
POWER_LAW, THERMAL_JUETTNER, PITCHY_PL, PITCHY_KAPPA = 0, 1, 2, 3

CONFIGS = {
    # name: (kind, seed offset, coefficient mask)
    "cfg2_powerlaw_jI_aI": (POWER_LAW, 0, 0x03),
    "cfg2_powerlaw_8": (POWER_LAW, 0, 0xFF),
    "cfg3_thermal_8": (THERMAL_JUETTNER, 1, 0xFF),
    "cfg4_pitchypl_8": (PITCHY_PL, 2, 0xFF),
    "cfg5_pitchykappa_8": (PITCHY_KAPPA, 3, 0xFF),
    # two variants SURVEY.md section 8d asks to be reported separately: the small-angle corner the tables leave out
    # (theta log-uniform in [1e-3, 0.05]: harmonics up to 1e12 and beyond), and gamma_min fixed at 1 as in the
    # reference's golden file (tests/symphony.rs:54-60) instead of log-uniform in [1, 30]
    "cfg2_powerlaw_8_corner": (POWER_LAW, 0, 0xFF),
    "cfg2_powerlaw_8_gmin1": (POWER_LAW, 0, 0xFF),
}


def make_batch(config, n, start=0):
    """Return (kind, mask, s, theta, [params...]) for points start .. start+n-1 of a config."""
    return make_rows(config, np.arange(start, start + n, dtype=np.uint64))


def make_rows(config, rows):
    """The same for an arbitrary set of row indices of a config's table (the generator is counter-based: row i is a
    function of i alone) -- e.g. one rank's interleaved share rank, rank + world, ... of a table."""
    kind, off, mask = CONFIGS[config]
    seed = SEED_BASE + off
    idx = np.asarray(rows, dtype=np.uint64)
    n = len(idx)
    s = _log(uniform01(seed, idx, 0), 0.1, 1e4)
    theta = _lin(uniform01(seed, idx, 1), 0.05, 1.52)
    if config.endswith("_corner"):
        theta = _log(uniform01(seed, idx, 1), 1e-3, 0.05)
    ones = np.ones(n)
    if kind == POWER_LAW:
        p = _lin(uniform01(seed, idx, 2), 1.5, 4.0)
        gmin = _log(uniform01(seed, idx, 3), 1.0, 30.0)
        if config.endswith("_gmin1"):
            gmin = ones.copy()
        params = [p, gmin, 1e12 * ones, 1e10 * ones]
    elif kind == THERMAL_JUETTNER:
        params = [_log(uniform01(seed, idx, 2), 0.1, 100.0)]
    elif kind == PITCHY_PL:
        p = _lin(uniform01(seed, idx, 2), 1.5, 4.0)
        k = _lin(uniform01(seed, idx, 3), 0.0, 3.0)
        params = [p, k, ones.copy(), 1e12 * ones, 1e10 * ones]
    else:
        kappa = _lin(uniform01(seed, idx, 2), 1.5, 4.5)
        width = _log(uniform01(seed, idx, 3), np.e, np.e ** 3)
        k = _lin(uniform01(seed, idx, 4), 0.0, 3.0)
        params = [kappa, width, k, 1e10 * ones]
    return kind, mask, s, theta, params


def compare_tables(got, ref, mask=0xFF):
    """Relative-error distribution of one [n][8] table against another over the slots selected in `mask`:
    median / p99 / max of |got - ref| / |ref| where both are finite, the fraction of coefficients within 1e-6, and
    how many coefficients are NaN on exactly one side (the NaN pattern is part of the result: a NaN is the
    reference's way of reporting a failed quadrature, symphony.rs:115-117, 380)."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    slots = [k for k in range(8) if mask & (1 << k)]
    g, r = got[:, slots], ref[:, slots]
    both = np.isfinite(g) & np.isfinite(r)
    with np.errstate(divide="ignore", invalid="ignore"):
        rel = np.abs(g - r) / np.abs(r)
    rel = np.where(r == g, 0.0, rel)[both]
    res = {"rows": int(got.shape[0]), "coefficients": int(g.size), "both_finite": int(both.sum()),
           "nan_only_here": int((~np.isfinite(g) & np.isfinite(r)).sum()),
           "nan_only_there": int((np.isfinite(g) & ~np.isfinite(r)).sum()),
           "nan_both": int((~np.isfinite(g) & ~np.isfinite(r)).sum())}
    if rel.size:
        res.update({"median": float(np.median(rel)), "p99": float(np.quantile(rel, 0.99)), "max": float(rel.max()),
                    "within_1e-6": float((rel <= 1e-6).mean()), "bit_identical": float((rel == 0).mean())})
    return res
