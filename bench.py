#!/usr/bin/env python3
"""bench.py -- throughput of the batched synchrotron-coefficient hot path on MI355X.

Primary workload (`value`, BASELINE.json configs[1]): power-law distribution, the 1e6-point synthetic table of
random (s, theta, p, gamma_min) of rimphony_amd/workload.py, coefficients j_I and alpha_I, fp64.  A "step" is one
pass of the hot path (full_calculation + the selected coefficients) over `--points` consecutive table rows PER GPU
(default 262144); successive steps walk through the table.  With N GPUs the step's N*points rows are sharded
interleaved (row i -> rank i mod N, no data-path collective during compute) and the output table is gathered to
rank 0 with one RCCL gather inside the timed region.  Inputs are resident in HBM before the timed region starts.

`python bench.py --gpus N` launches itself: without a launcher's WORLD_SIZE in the environment the parent starts
N rank processes (rimphony_amd/launch.py) before anything has touched the GPU and exits with their status; under
`python -m torch.distributed.run ... bench.py --gpus N` the ranks are the launcher's.

Prints ONE JSON line (rank 0).  Besides the contract's fields:
  roofline       the dominant kernel of the primary workload (coop_kernel<SymphonyProblem<0>>): modelled fp64 flops
                 (device-counted integrand samples x the hand-counted flops per sample of DESIGN.md) over its
                 HIP-event duration, against the fp64 VECTOR peak -- the path is VALU-issue-bound, not HBM- or
                 MFMA-bound, hence "bound": "valu_fp64".
  eight_coeff    BASELINE.json's metric proper: all eight coefficients per point (power law), timed over
                 EIGHT_STEPS steps of EIGHT_ROWS rows per GPU, with a roofline object for BOTH kernels.
  thermal_eight  the same on configs[2]'s table (thermal Juettner, the largest single-GPU configuration).
  parity         HIP output against the committed vectors of the oracle's LITERAL flavour (tests/golden/literal_*.npz:
                 glibc libm, unfused, GSL summation order): median / p99 / max relative error and NaN-pattern
                 mismatches -- the stand-in for BASELINE's "max rel-err vs Rust/GSL ref" (the HIP path is bit-identical
                 to the oracle's deterministic flavour; tests/test_gpu_parity.py).
  cpu_baseline   the oracle (a port, not the Rust binary) on the host cores over a bounded prefix of the table.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# DESIGN.md "Algorithmic work per unit": fp64 flops per integrand sample (FMA = 2; add, mul, div, sqrt = 1; elementary
# functions expanded), counted from the source and weighted with the measured branch mix of each table.  These are
# MODELS of the algorithmic work, not counter readings; the executed instruction mix is in profiles/r2_pmc_*.json.
SYMPHONY_FLOPS = {"cfg2_powerlaw_jI_aI": 720.0, "cfg2_powerlaw_8": 720.0, "cfg3_thermal_8": 600.0}
FARADAY_FLOPS = {"cfg2_powerlaw_8": 498.0, "cfg3_thermal_8": 617.0}      # profiles/r2_faraday_flop_model.txt
FP64_VECTOR_PEAK_TFLOPS = 78.6      # MI355X public spec, 256 CUs x 128 flop/clk x 2.4 GHz
# HBM-side bytes per integrand sample of the symphony kernel, from rocprofv3 PMC (FETCH_SIZE and WRITE_SIZE, separate
# passes, KB -> bytes; narrow accesses, so the gfx950 "wide read" doubling does not apply) on one 65536-row launch.
PMC_PROFILE = "profiles/r2_pmc_symphony_65536pts.json"
PMC_BYTES_PER_SAMPLE = (6.920e6 + 1.048e8) * 1024. / 34183156539.     # FETCH_SIZE + WRITE_SIZE (KB) / samples of that launch
TABLE = 1_000_000
EIGHT_ROWS, EIGHT_STEPS = 65536, 5
THERMAL_ROWS, THERMAL_STEPS = 65536, 2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--points", type=int, default=262144, help="table rows per GPU per step")
    ap.add_argument("--config", default="cfg2_powerlaw_jI_aI")
    ap.add_argument("--cpu-sample", type=int, default=2048, help="points of the CPU baseline sample (0 = skip)")
    ap.add_argument("--eight-rows", type=int, default=EIGHT_ROWS,
                    help="rows per GPU per step of the eight-coefficient legs (0 = skip them)")
    ap.add_argument("--no-parity", action="store_true", help="skip the comparison with the literal-oracle vectors")
    args = ap.parse_args()

    from rimphony_amd import launch
    if args.gpus > 1 and not launch.launched_by_launcher():
        # no GPU call has been made in this process (torch is not even imported): start the ranks and leave
        sys.exit(launch.spawn_ranks(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist
    from rimphony_amd import api, sharding, workload

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started %d ranks" % (args.gpus, world))
    distributed = world > 1
    # Rehearsal switch for 1-GPU boxes: RIMPHONY_BENCH_REHEARSE=1 puts every rank on cuda:0 and uses gloo
    # (the gather then goes through host memory).  The real multi-GPU run uses RCCL ("nccl"), one GPU per rank.
    rehearse = os.environ.get("RIMPHONY_BENCH_REHEARSE") == "1"
    # one GPU per rank; if a launcher narrowed the visible devices to one per rank, that one is index 0
    dev_index = 0 if rehearse else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    if distributed:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))

    ctx = api.Context(dev_index)
    dev = torch.device("cuda", dev_index)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(x):
        if not distributed:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def stage(config, rows, start):
        """This rank's interleaved shard of `rows * world` table rows, resident in HBM."""
        kind, mask, s, th, params = workload.make_batch(config, rows * world, start=start)
        mine = sharding.shard_indices(rows * world, rank, world)
        return kind, mask, (torch.from_numpy(s[mine]).to(dev), torch.from_numpy(th[mine]).to(dev),
                            [torch.from_numpy(p[mine]).to(dev) for p in params])

    def timed_leg(config, rows, steps, warm_rows, start0, wrap):
        """`steps` timed steps of `rows` rows per GPU (after one untimed step of warm_rows rows); returns the dict
        of whole-job rate, per-kernel HIP-event times and device work counters (rank-local kernel figures)."""
        shards = []
        for st in range(steps):
            start = start0 + (st * rows * world) % wrap
            shards.append(stage(config, rows, start))
        kind, mask, w = stage(config, warm_rows, start0)
        ctx.compute_batch_device(kind, w[0], w[1], w[2], mask)
        sym_ms, far_ms, sym_samples, far_samples = [], [], [], []
        barrier()
        t0 = time.perf_counter()
        for kind, mask, d in shards:
            out, _ = ctx.compute_batch_device(kind, d[0], d[1], d[2], mask)
            if distributed:
                sharding.gather_table(out.cpu() if rehearse else out, rows * world, rank, world, dst=0)
            if mask & 0x3F:
                sym_ms.append(ctx.last_symphony_ms())          # HIP events on the launch stream
            if mask & 0xC0:
                far_ms.append(ctx.last_faraday_ms())
            wk = ctx.last_work()
            sym_samples.append(wk["samples"])
            far_samples.append(wk["faraday_samples"])
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        return {"dt": dt, "sym_ms": sym_ms, "far_ms": far_ms, "sym_samples": sym_samples, "far_samples": far_samples}

    def roofline(kernel, flops_per_sample, ms, samples, with_traffic=False):
        avg_s = float(np.mean(ms)) * 1e-3
        avg_samples = float(np.mean(samples))
        achieved = avg_samples * flops_per_sample / avg_s / 1e12
        r = {"bound": "valu_fp64", "achieved": round(achieved, 4), "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
             "frac": round(achieved / FP64_VECTOR_PEAK_TFLOPS, 5), "traffic": None,
             "kernel": kernel, "kernel_ms": round(avg_s * 1e3, 3), "samples_per_launch": avg_samples,
             "flops_per_sample": flops_per_sample,
             "flops_kind": "modelled: hand-counted algorithmic flops per integrand sample (DESIGN.md section 5) x "
                           "samples counted on the device in this run"}
        if with_traffic:
            r["traffic"] = round(PMC_BYTES_PER_SAMPLE * avg_samples)
            r["traffic_kind"] = "extrapolated: PMC bytes per sample of %s x samples of this run" % PMC_PROFILE
        return r

    # ---------------------------------------------------------------- primary: configs[1], timed per the contract
    P = args.points
    kind, mask, _, _, _ = workload.make_batch(args.config, 1)
    nsel = bin(mask & 0xFF).count("1")
    total_steps = args.warmup + args.steps
    shards = [stage(args.config, P, (st * P * world) % TABLE)[2] for st in range(total_steps)]
    kernel_ms, samples = [], []

    def run_step(i, record):
        s, th, params = shards[i]
        out, _ = ctx.compute_batch_device(kind, s, th, params, mask)
        if distributed:
            # RCCL gather of the output table (host tensors in the gloo rehearsal)
            sharding.gather_table(out.cpu() if rehearse else out, P * world, rank, world, dst=0)
        if record:
            kernel_ms.append(ctx.last_symphony_ms())        # HIP events on the launch stream
            samples.append(ctx.last_work()["samples"])
        return out

    for i in range(args.warmup):
        run_step(i, False)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.warmup, total_steps):
        run_step(i, True)
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    del shards

    # ---------------------------------------------------------------- BASELINE's metric: eight coefficients per point
    eight = thermal = None
    if args.eight_rows > 0:
        R = args.eight_rows
        for name, cfg, rows, steps, start0 in (("eight", "cfg2_powerlaw_8", R, EIGHT_STEPS, TABLE),
                                               ("thermal", "cfg3_thermal_8", min(R, THERMAL_ROWS), THERMAL_STEPS, 0)):
            leg = timed_leg(cfg, rows, steps, min(rows, 4096), start0, TABLE)
            obj = {"value": round(rows * world * steps / leg["dt"], 2), "unit": "points/s", "coefficients_per_point": 8,
                   "rows_per_gpu_per_step": rows, "steps": steps, "ms_per_step": round(leg["dt"] / steps * 1e3, 3),
                   "workload": "%s, all 8 slots, rows %d.. of the generator" % (cfg, start0),
                   "roofline_symphony": roofline("coop_kernel<SymphonyProblem<%d>>" % workload.CONFIGS[cfg][0],
                                                 SYMPHONY_FLOPS[cfg], leg["sym_ms"], leg["sym_samples"]),
                   "roofline_faraday": roofline("coop_kernel<HeyvaertsProblem<%d>>" % workload.CONFIGS[cfg][0],
                                                FARADAY_FLOPS[cfg], leg["far_ms"], leg["far_samples"])}
            if name == "eight":
                eight = obj
            else:
                thermal = obj

    # ---------------------------------------------------------------- parity vs the literal-flavour vectors (rank 0)
    parity = None
    if rank == 0 and not args.no_parity:
        parity = {}
        for cfg in ("cfg2_powerlaw_jI_aI", "cfg2_powerlaw_8", "cfg3_thermal_8", "cfg4_pitchypl_8", "cfg5_pitchykappa_8"):
            path = os.path.join(ROOT, "tests", "golden", "literal_%s.npz" % cfg)
            if not os.path.exists(path):
                continue
            z = np.load(path)
            n, start, m = int(z["n"]), int(z["start"]), int(z["mask"])
            k2, _, s2, th2, p2 = workload.make_batch(cfg, n, start=start)
            got = ctx.compute_batch(k2, s2, th2, p2, m)
            parity[cfg] = workload.compare_tables(got, z["out"], m)
        parity["reference"] = ("tests/golden/literal_*.npz = oracle/liboracle_libm.so (glibc libm, unfused, GSL summation "
                               "order); made by tools/make_literal_fixtures.py")

    if rank == 0:
        points = P * world * args.steps
        value = points / elapsed
        roof = roofline("coop_kernel<SymphonyProblem<0>>", SYMPHONY_FLOPS[args.config] if args.config in SYMPHONY_FLOPS else 720.0,
                        kernel_ms, samples, with_traffic=True)

        cpu = None
        if args.cpu_sample > 0 and world == 1:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_bind
            L = oracle_bind.load("det")
            cores = min(len(os.sched_getaffinity(0)), 16)   # a 1-GPU box's CPU share is 16 cores
            _, _, s, th, params = workload.make_batch(args.config, args.cpu_sample, start=0)
            c0 = time.perf_counter()
            oracle_bind.batch(L, kind, s, th, params, mask, nthreads=cores)
            cdt = time.perf_counter() - c0
            cpu = {"value": round(args.cpu_sample / cdt, 3), "unit": "points/s", "cores": cores, "kind": "port",
                   "sample": "first %d rows of the same table, oracle/liboracle.so, OpenMP dynamic over points"
                             % args.cpu_sample}

        line = {
            "metric": "parameter-points/sec (power law, j_I + alpha_I per point)",
            "value": round(value, 2), "unit": "points/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: power_law, 1e6-row random (s,theta,p,gamma_min) table, "
                                   "j_I/alpha_I, fp64; step = %d rows per GPU" % P,
                       "points_per_step_per_gpu": P, "coefficients_per_point": nsel,
                       "coefficients_per_s": round(value * nsel, 2), "sharding": "interleaved, gather to rank 0"},
            "roofline": roof, "cpu_baseline": cpu, "eight_coeff": eight, "thermal_eight": thermal, "parity": parity,
        }
        print(json.dumps(line))

    ctx.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
