#!/usr/bin/env python3
"""bench.py -- throughput of the batched synchrotron-coefficient hot path on MI355X.

Workload (BASELINE.json configs[1]): power-law distribution, the 1e6-point
synthetic table of random (s, theta, p, gamma_min) defined in
rimphony_amd/workload.py, coefficients j_I and alpha_I, fp64.  A "step" is one
pass of the hot path (full_calculation + the selected coefficients) over one
batch of `--points` consecutive table rows PER GPU (default 262144, i.e. about a
quarter of the table per launch); successive steps walk through the table.  The
table contains rare points that are 100-1000x the mean cost (the reference would
print "SLOW" for them, tests/symphony.rs:63-67); the kernel's cooperative tail
 spreads such a point over the idle waves at the end of a
launch, so a launch with one costs ~0.2 s more than one without.  With N GPUs the step's global batch of N*points rows is
sharded interleaved (row i -> rank i mod N, no data-path collective during
compute) and the output table is gathered to rank 0 with one RCCL gather inside
the timed region.  Inputs are resident in HBM before the timed region starts.

Prints ONE JSON line (rank 0).  `value` = parameter points per second, whole job.
`roofline` is for the dominant kernel (coop_kernel<SymphonyProblem<0>>, "symphony kernel" below): algorithmic fp64 flops
(device-counted integrand samples x the per-sample figure of DESIGN.md) over
its HIP-event-measured duration, against the fp64 vector peak -- the path is
VALU-bound, not HBM- or MFMA-bound, hence "bound": "valu_fp64".
`cpu_baseline` times the oracle (a port, not the Rust binary) on the host cores
over a bounded prefix of the same table.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# DESIGN.md "Algorithmic work per unit": fp64 flops per integrand sample (power-law
# distribution; FMA = 2, add/mul/div/sqrt = 1, elementary functions expanded), counted from
# the source and weighted by the measured Bessel-region mix of this table
# (Debye 21 %, Meissel-1 73 %, blend 5 %, integer order 1 %).
FLOPS_PER_SAMPLE = 720.0
FP64_VECTOR_PEAK_TFLOPS = 78.6      # MI355X public spec, 256 CUs x 128 flop/clk x 2.4 GHz
# HBM-side bytes of the symphony kernel measured with rocprofv3 PMC (FETCH_SIZE and WRITE_SIZE, separate passes,
# KB -> bytes; narrow accesses, so the gfx950 "wide read" doubling does not apply) on one 65536-row launch of
# this table: profiles/r1_final_pmc_symphony_65536pts.json.  Nearly all of it is scratch (register spill)
# traffic; it scales with the sample count, hence the per-sample figure.  Algorithmic bytes are ~150 B/point.
PMC_BYTES_PER_SAMPLE = (6.504e5 + 3.5443e7) * 1024. / 34183156539.


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--points", type=int, default=262144, help="table rows per GPU per step")
    ap.add_argument("--config", default="cfg2_powerlaw_jI_aI")
    ap.add_argument("--cpu-sample", type=int, default=2048, help="points of the CPU baseline sample (0 = skip)")
    ap.add_argument("--eight-rows", type=int, default=16384,
                    help="rows per GPU of the secondary eight-coefficient step (0 = skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from rimphony_amd import api, sharding, workload

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
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
    kind, mask, _, _, _ = workload.make_batch(args.config, 1)
    nsel = bin(mask & 0x3F).count("1")

    P = args.points
    total_steps = args.warmup + args.steps
    TABLE = 1_000_000

    # stage every step's shard in HBM up front (inputs resident before timing)
    shards = []
    for st in range(total_steps):
        start = (st * P * world) % TABLE
        _, _, s, th, params = workload.make_batch(args.config, P * world, start=start)
        mine = sharding.shard_indices(P * world, rank, world)     # interleaved: row i -> rank i mod world
        shards.append((torch.from_numpy(s[mine]).to(dev), torch.from_numpy(th[mine]).to(dev),
                       [torch.from_numpy(p[mine]).to(dev) for p in params]))

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    kernel_ms = []
    samples = []

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
    elapsed = time.perf_counter() - t0

    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # Secondary figure, outside the timed region above: BASELINE.json words its metric as EIGHT-coefficient
    # points/s, while configs[1] selects two.  One step of the same table with all eight slots selected (the six
    # Symphony coefficients and the Faraday pair), --eight-rows rows per GPU, same sharding, gather and timing rules.
    eight = None
    if args.eight_rows > 0:
        R = args.eight_rows
        k8, m8, s8, th8, p8 = workload.make_batch("cfg2_powerlaw_8", R * world, start=TABLE)
        mine = sharding.shard_indices(R * world, rank, world)
        d8 = (torch.from_numpy(s8[mine]).to(dev), torch.from_numpy(th8[mine]).to(dev),
              [torch.from_numpy(q[mine]).to(dev) for q in p8])

        def eight_step(rows):
            out8, _ = ctx.compute_batch_device(k8, d8[0][:rows], d8[1][:rows], [q[:rows] for q in d8[2]], m8)
            if distributed and rows == R:
                sharding.gather_table(out8.cpu() if rehearse else out8, R * world, rank, world, dst=0)

        eight_step(256)                     # loads the Faraday kernel and its series table
        barrier()
        e0 = time.perf_counter()
        eight_step(R)
        barrier()
        e_dt = time.perf_counter() - e0
        if distributed:
            t = torch.tensor([e_dt], dtype=torch.float64, device="cpu" if rehearse else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e_dt = float(t.item())
        eight = {"value": round(R * world / e_dt, 2), "unit": "points/s", "rows_per_gpu": R,
                 "ms": round(e_dt * 1e3, 3), "symphony_kernel_ms": round(ctx.last_symphony_ms(), 3),
                 "faraday_kernel_ms": round(ctx.last_faraday_ms(), 3),
                 "note": "one step, all 8 slots (power law), rows %d.. of the same generator" % TABLE}

    if rank == 0:
        points = P * world * args.steps
        value = points / elapsed
        avg_kernel_s = float(np.mean(kernel_ms)) * 1e-3
        avg_samples = float(np.mean(samples))
        achieved = avg_samples * FLOPS_PER_SAMPLE / avg_kernel_s / 1e12
        roofline = {
            "bound": "valu_fp64", "achieved": round(achieved, 4), "peak": FP64_VECTOR_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": round(achieved / FP64_VECTOR_PEAK_TFLOPS, 5),
            "traffic": round(PMC_BYTES_PER_SAMPLE * avg_samples),
            "kernel": "coop_kernel<SymphonyProblem<0>>", "kernel_ms": round(avg_kernel_s * 1e3, 3),
            "samples_per_launch": avg_samples, "flops_per_sample": FLOPS_PER_SAMPLE,
            "traffic_note": "bytes/launch = PMC bytes/sample (profiles/r1_final_pmc_symphony_65536pts.json) x samples",
        }

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
            "roofline": roofline, "cpu_baseline": cpu, "eight_coeff": eight,
        }
        print(json.dumps(line))

    ctx.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
