"""Batched crank-out drivers: the neurosynchro training-set generators of the reference
(examples/crank-out-pitchypl.rs, examples/crank-out-pitchykappa.rs) on top of the batched GPU
path.  Same command-line arguments, same TSV format (header line re-written on every start,
append mode, `{:.16e}` numbers in Rust's spelling), but parameters are drawn and evaluated a
block at a time instead of one `compute_all_dimensionless` call per loop turn.

Differences that the format cannot hide, stated here rather than glossed over:
  * `time_ms(meta)` (crank-out-pitchypl.rs:167-173 times each point's `compute_all_dimensionless` on its CPU): a
    block is one launch, so a row's time is its SHARE of the block's two kernel times, apportioned by the integrand
    samples the device counted for that row's coefficients (the `work` array of rimphony_batch_compute_ex: Symphony
    samples against the Symphony kernel's time, Faraday samples against the Faraday kernel's) -- a per-row cost that
    ranks rows as the reference's column does and sums to the time the block took;
  * the pitchy-kappa driver of the reference writes and flushes the parameters of a point BEFORE
    computing it, so a hang can be reconstructed (crank-out-pitchykappa.rs:193-200).  Here the
    block's parameters are written, flushed to `<OUTFILE>.pending` before the launch and the file
    is removed once the block's rows are in OUTFILE; together with `rimphony_debug_heartbeat` that
    serves the same purpose.
  * the reference loops forever; `--count` bounds the run (0 = forever);
  * `--gpus N`: one context per GPU in this process, each block sharded over them by the library's multi-device
    entry (rimphony_batch_compute_multi: row i -> GPU i mod N); the rows of a block do not depend on N.

Usage:
  python -m rimphony_amd.crank_out pitchypl  OUTFILE S_MIN S_MAX THETA_MIN THETA_MAX P_MIN P_MAX K_MIN K_MAX
  python -m rimphony_amd.crank_out pitchykappa OUTFILE S_MIN S_MAX THETA_MIN THETA_MAX KAPPA_MIN KAPPA_MAX \
                                   WIDTH_MIN WIDTH_MAX K_MIN K_MAX
"""
import argparse
import math
import os
import sys
import time

import numpy as np

HEADER_PITCHYPL = ("s(log)\ttheta(lin)\tp(lin)\tk(lin)\ttime_ms(meta)\tj_I(res)\talpha_I(res)\tj_Q(res)\t"
                   "alpha_Q(res)\tj_V(res)\talpha_V(res)\trho_Q(res)\trho_V(res)")
HEADER_PITCHYKAPPA = ("s(log)\ttheta(lin)\tkappa(lin)\twidth(log)\tk(lin)\ttime_ms(meta)\tj_I(res)\talpha_I(res)\t"
                      "j_Q(res)\talpha_Q(res)\tj_V(res)\talpha_V(res)\trho_Q(res)\trho_V(res)")


def rust_e16(x):
    """Rust's `{:.16e}` for an f64: 16 fractional digits, exponent without '+' or zero padding."""
    x = float(x)
    if math.isnan(x):
        return "NaN"
    if math.isinf(x):
        return "inf" if x > 0 else "-inf"
    mant, exp = ("%.16e" % x).split("e")
    return "%se%d" % (mant, int(exp))


class Sampler:
    """test-support/src/lib.rs:31-64: uniform, or log-uniform via exp(U[ln lo, ln hi])."""

    def __init__(self, is_log, low, high, rng):
        if low > high:
            low, high = high, low
        if is_log:
            low, high = math.log(low), math.log(high)
        self.is_log, self.low, self.range, self.rng = is_log, low, high - low, rng

    def get(self, n):
        v = self.low + self.rng.random(n) * self.range
        return np.exp(v) if self.is_log else v


def format_rows(param_cols, ms, vals):
    """ms: one value for the block or one per row."""
    ms = np.broadcast_to(np.asarray(ms, dtype=np.float64), (vals.shape[0],))
    lines = []
    for i in range(vals.shape[0]):
        fields = [rust_e16(c[i]) for c in param_cols] + [rust_e16(ms[i])] + [rust_e16(v) for v in vals[i]]
        lines.append("\t".join(fields))
    return lines


def row_times_ms(work, symphony_ms, faraday_ms):
    """A row's share of the block's kernel times: its Symphony samples of the Symphony kernel's time plus its Faraday
    samples of the Faraday kernel's time (work: [n, 8] integrand samples per coefficient)."""
    work = np.asarray(work, dtype=np.float64)
    sym, far = work[:, :6].sum(axis=1), work[:, 6:].sum(axis=1)
    ms = np.zeros(work.shape[0])
    if sym.sum() > 0:
        ms += symphony_ms * sym / sym.sum()
    if far.sum() > 0:
        ms += faraday_ms * far / far.sum()
    return ms


def run(kind_name, outfile, ranges, count, block, seed, compute):
    """compute(kind, s, theta, params) -> [n, 8] array, or (that array, per-row milliseconds).  Returns the number
    of rows written."""
    from . import workload
    rng = np.random.default_rng(seed)
    if kind_name == "pitchypl":
        header = HEADER_PITCHYPL
        samplers = [Sampler(True, *ranges[0:2], rng), Sampler(False, *ranges[2:4], rng),
                    Sampler(False, *ranges[4:6], rng), Sampler(False, *ranges[6:8], rng)]
    else:
        header = HEADER_PITCHYKAPPA
        samplers = [Sampler(True, *ranges[0:2], rng), Sampler(False, *ranges[2:4], rng),
                    Sampler(False, *ranges[4:6], rng), Sampler(True, *ranges[6:8], rng),
                    Sampler(False, *ranges[8:10], rng)]
    written = 0
    pending = outfile + ".pending"
    with open(outfile, "a") as f:
        f.write(header + "\n")
        while count == 0 or written < count:
            n = block if count == 0 else min(block, count - written)
            cols = [smp.get(n) for smp in samplers]
            s, theta = cols[0], cols[1]
            if kind_name == "pitchypl":
                p, k = cols[2], cols[3]
                kind = workload.PITCHY_PL
                params = [p, k, np.ones(n), 1e12 * np.ones(n), 1e10 * np.ones(n)]   # crank-out-pitchypl.rs:163-165
            else:
                kappa, width, k = cols[2], cols[3], cols[4]
                kind = workload.PITCHY_KAPPA
                params = [kappa, width, k, 1e10 * np.ones(n)]                          # crank-out-pitchykappa.rs:191
            with open(pending, "w") as pf:       # parameters before results
                for i in range(n):
                    pf.write("\t".join(rust_e16(c[i]) for c in cols) + "\n")
                pf.flush()
                os.fsync(pf.fileno())
            t0 = time.perf_counter()
            res = compute(kind, s, theta, params)
            if isinstance(res, tuple):
                vals, ms = res
            else:                                # a compute without per-row costs: the block's mean
                vals, ms = res, (time.perf_counter() - t0) * 1e3 / n
            f.write("\n".join(format_rows(cols, ms, vals)) + "\n")
            f.flush()
            os.remove(pending)
            written += n
    return written


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("kind", choices=["pitchypl", "pitchykappa"])
    ap.add_argument("outfile")
    ap.add_argument("ranges", nargs="+", type=float, help="MIN MAX pairs in the order of the reference driver")
    ap.add_argument("--count", type=int, default=0, help="rows to write (0 = run forever, like the reference)")
    # the end of a launch (its longest chains of batches) takes about as long whatever the block's size: large blocks amortise
    # it -- pitchy kappa: 6.8 k rows/s in 16384-row blocks, 11.8 k rows/s in 625000-row ones (profiles/r4_full_size_*.txt)
    ap.add_argument("--block", type=int, default=262144, help="rows per launch (per GPU with --gpus)")
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--gpus", type=int, default=1, help="GPUs of this node to shard each block over")
    args = ap.parse_args(argv)
    need = 8 if args.kind == "pitchypl" else 10
    if len(args.ranges) != need:
        ap.error("%s takes %d range values" % (args.kind, need))
    from . import api
    ctxs = [api.Context(d) for d in range(args.gpus)]
    run(args.kind, args.outfile, args.ranges, args.count, args.block, args.seed, gpu_compute(ctxs))
    for c in ctxs:
        c.close()


def gpu_compute(ctxs):
    """The driver's compute callback on one or several GPU contexts, with per-row times from the work counters."""
    from . import api

    def compute(kind, s, th, params):
        t0 = time.perf_counter()
        if len(ctxs) == 1:
            vals, work = ctxs[0].compute_batch(kind, s, th, params, api.SLOTS_ALL, want_work=True)
            return vals, row_times_ms(work, ctxs[0].last_symphony_ms(), ctxs[0].last_faraday_ms())
        vals, work = api.compute_batch_multi(ctxs, kind, s, th, params, api.SLOTS_ALL, want_work=True)
        # several devices ran side by side: apportion the block's wall time by total samples
        ms = (time.perf_counter() - t0) * 1e3
        w = np.asarray(work, dtype=np.float64).sum(axis=1)
        return vals, ms * w / max(w.sum(), 1.)
    return compute


if __name__ == "__main__":
    main()
