#!/usr/bin/env python3
"""Benchmark of the hot path: eight-coefficient parameter-points per second (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path, N x (full_calculation + compute_all_dimensionless) with ALL EIGHT coefficients per
point (lib.rs:178-191), over one batch of `--points` synthetic power-law points per GPU (the generator of
rimphony_amd/workload.py, rows 1000000.. of its table), inputs resident in HBM before the timed region.  For N > 1 the
batch is sharded interleaved over the ranks (one process per GPU) and the per-rank tables are gathered to rank 0 with
one RCCL gather per step; `value` = points of all ranks / max-over-ranks time, "scaling": "weak".

Side objects of the JSON line:
  roofline          the dominant kernel of a step (group_kernel<0>: the six Symphony coefficients of a point in
                    lock-step): algorithmic flops = device-counted integrand samples (the REFERENCE's count: every
                    coefficient's own samples, whether or not the kernel shared their evaluation) x the hand-counted flops
                    per sample, over the kernel's HIP-event time, against the fp64 vector peak; `traffic` = HBM bytes per
                    launch from the committed PMC profile of THIS build (profiles/r4_pmc_*.json carries the sample count of
                    the profiled launch and the source id of the build; a stale profile is refused, traffic = null);
                    `executed` = the fp64 work the vector unit really did (PMC instruction counts per pass x this run's
                    passes): the hardware-utilisation figure, below `achieved` where samples are shared
  roofline_faraday  the same for coop_kernel<HeyvaertsProblem<0>> (rho_Q, rho_V)
  two_coeff         BASELINE configs[1] (power law, j_I + alpha_I only), with its own roofline
  thermal_eight     configs[2]'s table (thermal Juettner, eight coefficients)
  pitchypl_eight, pitchykappa_eight   configs[3] / configs[4]'s tables (the anisotropic distributions), with both rooflines
  corner / gmin1    SURVEY 8d's two separately-reported variants: theta < 0.05, and gamma_min = 1 as in the golden file
  parity            HIP output against the committed vectors of the oracle's LITERAL flavour (tests/golden/literal_*.npz:
                    glibc libm, unfused, GSL summation order) -- the stand-in for BASELINE's "max rel-err vs Rust/GSL ref"
                    -- and `control`: the distance between the literal flavour and two equally legitimate builds of it
                    (GK31 terms added in reverse order; -ffp-contract=fast): the noise floor of the reference's arithmetic
  cpu_baseline      the oracle (a port, not the Rust binary) on the host cores over bounded prefixes of the same tables:
                    {1 core, all cores} x {2 coefficients, 8 coefficients}, with the CPU model string
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
# MODELS of the algorithmic work, not counter readings; the executed instruction mix is in profiles/r3_pmc_*.json.
# The anisotropic tables: the power-law sample (720) plus the pitch-angle factor sin^k xi -- one sqrt, one pow (90) and
# three products, + d f / d cos xi on the absorption side (pitchy_pl.rs:32-64) = 816; pitchy kappa: the thermal sample
# (600) with exp (34 + 8) replaced by the kappa power and the cutoff exponential (pow 90 + exp 34 + 4), the same
# pitch-angle factor (94) and the derivative's quotient (8) (pitchy_kappa.rs:38-62) = 788.
SYMPHONY_FLOPS = {"cfg2_powerlaw_jI_aI": 720.0, "cfg2_powerlaw_8": 720.0, "cfg3_thermal_8": 600.0,
                  "cfg4_pitchypl_8": 816.0, "cfg5_pitchykappa_8": 788.0}
# tools/faraday_flop_model.py (profiles/r2_faraday_flop_model.txt): hand count of the reference's elements x the
# oracle-measured branch mix of each table
FARADAY_FLOPS = {"cfg2_powerlaw_8": 498.0, "cfg3_thermal_8": 617.0, "cfg4_pitchypl_8": 902.0, "cfg5_pitchykappa_8": 1081.0}
FP64_VECTOR_PEAK_TFLOPS = 78.6      # MI355X public spec, 256 CUs x 128 flop/clk x 2.4 GHz
# committed counter profiles of ONE launch of each persistent kernel on the power-law table (tools/pmc_collect.sh):
# HBM bytes and executed fp64 instructions per sample / per pass, quoted only for the build they were taken on
PMC_PROFILES = {"group_kernel": "profiles/r4_pmc_group_powerlaw8.json", "Heyvaerts": "profiles/r4_pmc_faraday_powerlaw8.json"}
TABLE = 1_000_000
REFERENCE_HINT = ("the reference's only timing statement: benches/powerlaw.rs:6-8, 'about 40 minutes' for 16 single-coefficient "
                  "benchmarks x >= 300 iterations on its author's machine, i.e. about 0.5 s per coefficient on one core")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--points", type=int, default=131072, help="table rows per GPU per step")
    ap.add_argument("--config", default="cfg2_powerlaw_8")
    ap.add_argument("--cpu-sample", type=int, default=256,
                    help="rows of the all-cores eight-coefficient CPU baseline (the other three are scaled from it; 0 = skip)")
    ap.add_argument("--side-rows", type=int, default=65536, help="rows per GPU per step of the side legs (0 = skip them)")
    ap.add_argument("--no-parity", action="store_true", help="skip the comparison with the literal-oracle vectors")
    args = ap.parse_args()

    from rimphony_amd import launch
    if args.gpus > 1 and not launch.launched_by_launcher():
        # no GPU call has been made in this process (torch is not even imported): start the ranks and leave
        sys.exit(launch.spawn_ranks(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist
    from rimphony_amd import _build, api, sharding, workload

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started %d ranks" % (args.gpus, world))
    # RIMPHONY_BENCH_FORCE_DIST=1: initialise RCCL and run the gather in a world of ONE as well (tests: the RCCL path
    # executed on a 1-GPU box)
    force_dist = os.environ.get("RIMPHONY_BENCH_FORCE_DIST") == "1"
    distributed = world > 1 or force_dist
    # Rehearsal switch for 1-GPU boxes: RIMPHONY_BENCH_REHEARSE=1 puts every rank on cuda:0 and uses gloo
    # (the gather then goes through host memory).  The real multi-GPU run uses RCCL ("nccl"), one GPU per rank.
    rehearse = os.environ.get("RIMPHONY_BENCH_REHEARSE") == "1"
    # one GPU per rank; if a launcher narrowed the visible devices to one per rank, that one is index 0
    dev_index = 0 if rehearse else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    if distributed:
        if force_dist and "MASTER_ADDR" not in os.environ:
            os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(launch.free_port()), "RANK": "0", "WORLD_SIZE": "1"})
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

    def minmax_over_ranks(x):
        if not distributed:
            return [x, x]
        t = torch.tensor([x, -x], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(-t[1].item()), float(t[0].item())]

    def stage(config, rows, start):
        """This rank's interleaved shard of `rows * world` table rows, resident in HBM."""
        kind, mask, s, th, params = workload.make_batch(config, rows * world, start=start)
        mine = sharding.shard_indices(rows * world, rank, world)
        return kind, mask, (torch.from_numpy(s[mine]).to(dev), torch.from_numpy(th[mine]).to(dev),
                            [torch.from_numpy(p[mine]).to(dev) for p in params])

    def timed_leg(config, rows, steps, warmup, start0, wrap, mask_override=None):
        """`warmup` untimed + `steps` timed steps of `rows` rows per GPU; timed region bracketed by barrier + synchronize on
        both sides, max over ranks.  Returns the whole-job time, per-kernel HIP-event times and device work counters."""
        shards = [stage(config, rows, start0 + (st * rows * world) % wrap) for st in range(warmup + steps)]
        sym_ms, far_ms, sym_samples, sym_passes, far_samples, far_passes = [], [], [], [], [], []

        def run(i, record):
            kind, mask, d = shards[i]
            if mask_override is not None:
                mask = mask_override
            out, _ = ctx.compute_batch_device(kind, d[0], d[1], d[2], mask)
            if distributed:
                # RCCL gather of the output table (host tensors in the gloo rehearsal)
                sharding.gather_table(out.cpu() if rehearse else out, rows * world, rank, world, dst=0, force=force_dist)
            if record:
                if mask & 0x3F:
                    sym_ms.append(ctx.last_symphony_ms())          # HIP events on the launch stream
                if mask & 0xC0:
                    far_ms.append(ctx.last_faraday_ms())
                wk = ctx.last_work()
                sym_samples.append(wk["samples"])
                sym_passes.append(wk["passes"])
                far_samples.append(wk["faraday_samples"])
                far_passes.append(wk["faraday_passes"])

        for i in range(warmup):
            run(i, False)
        barrier()
        t0 = time.perf_counter()
        for i in range(warmup, warmup + steps):
            run(i, True)
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        return {"dt": dt, "sym_ms": sym_ms, "far_ms": far_ms, "sym_samples": sym_samples, "sym_passes": sym_passes,
                "far_samples": far_samples, "far_passes": far_passes}

    def pmc_record(kernel):
        """The committed PMC record of `kernel`'s family, or (None, why): only a profile of THIS build is quoted."""
        fam = next((f for f in PMC_PROFILES if f in kernel), None)
        if fam is None:
            return None, "no PMC profile for this kernel"
        path = os.path.join(ROOT, PMC_PROFILES[fam])
        if not os.path.exists(path):
            return None, "no PMC profile committed (%s)" % PMC_PROFILES[fam]
        rec = next((r for r in json.load(open(path)) if fam in r["kernel"]), None)
        if rec is None or "work" not in rec or not rec["work"].get("samples"):
            return None, "PMC profile without the launch's sample count"
        if rec.get("source_id") != _build.source_id():
            return None, "stale: %s was taken on source id %s, this build is %s" % (PMC_PROFILES[fam], rec.get("source_id"), _build.source_id())
        rec["path"] = PMC_PROFILES[fam]
        return rec, None

    def roofline(kernel, flops_per_sample, ms, samples, passes=None, with_traffic=False, kind=0):
        """`achieved` / `frac`: ALGORITHMIC work (the reference's samples x modelled flops per sample) over the kernel's
        HIP-event time -- the contract's definition; a kernel that shares samples between coefficients scores above
        its hardware utilisation here.  `executed`: what the vector unit actually did -- fp64 flop-lanes per executed
        pass from the committed PMC profile of this build (2 FMA + ADD + MUL + TRANS instructions x 64 lanes,
        masked-off lanes included) x the passes counted on the device in this run: the utilisation figure."""
        avg_s = float(np.mean(ms)) * 1e-3
        avg_samples = float(np.mean(samples))
        achieved = avg_samples * flops_per_sample / avg_s / 1e12
        r = {"bound": "valu_fp64", "achieved": round(achieved, 4), "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
             "frac": round(achieved / FP64_VECTOR_PEAK_TFLOPS, 5), "traffic": None,
             "kernel": kernel, "kernel_ms": round(avg_s * 1e3, 3), "samples_per_launch": avg_samples,
             "flops_per_sample": flops_per_sample,
             "flops_kind": "ALGORITHMIC: hand-counted flops per integrand sample of the reference's algorithm "
                           "(DESIGN.md section 5) x the reference's sample count, counted on the device in this run; "
                           "not a utilisation figure -- see `executed`"}
        if passes:
            # 62 samples per pass and coefficient in the reference's scheme; the group kernel serves several
            # coefficients with one pass
            r["samples_per_executed_pass"] = round(avg_samples / float(np.mean(passes)), 2)
        rec, why = pmc_record(kernel)
        if rec is not None and kind == 0:
            c = rec["counters"]
            if with_traffic:
                per_sample = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024. / rec["work"]["samples"]
                r["traffic"] = round(per_sample * avg_samples)
                r["traffic_kind"] = ("PMC FETCH_SIZE + WRITE_SIZE of %s (%d samples in that launch, same source id) scaled to "
                                     "the samples of this run's launch" % (rec["path"], rec["work"]["samples"]))
            if passes and rec["work"].get("passes"):
                lanes = 64. * (2. * c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"]
                               + c["SQ_INSTS_VALU_TRANS_F64"]) / rec["work"]["passes"]
                ex = lanes * float(np.mean(passes)) / avg_s / 1e12
                r["executed"] = {"tflops": round(ex, 3), "frac": round(ex / FP64_VECTOR_PEAK_TFLOPS, 5),
                                 "fp64_flop_lanes_per_pass": round(lanes, 1),
                                 "valu_per_pass": round(c["SQ_INSTS_VALU"] / rec["work"]["passes"], 1),
                                 "valu_busy": round(c["SQ_ACTIVE_INST_VALU"] * 4. / 1024. / (c["GRBM_GUI_ACTIVE"] / 8.), 4),
                                 "basis": "PMC instruction counts per executed pass of %s (same source id) x this run's passes" % rec["path"]}
        elif with_traffic:
            r["traffic_kind"] = why
        kms = minmax_over_ranks(avg_s * 1e3)
        if distributed:
            r["kernel_ms_min_max_over_ranks"] = [round(kms[0], 3), round(kms[1], 3)]
        return r

    def side_leg(cfg, rows, steps, start0, mask_override=None, with_roofline=True):
        leg = timed_leg(cfg, rows, steps, 1, start0, TABLE, mask_override)
        base = cfg.replace("_corner", "").replace("_gmin1", "")
        kind = workload.CONFIGS[cfg][0]
        obj = {"value": round(rows * world * steps / leg["dt"], 2), "unit": "points/s",
               "coefficients_per_point": bin((mask_override if mask_override is not None else workload.CONFIGS[cfg][2]) & 0xFF).count("1"),
               "rows_per_gpu_per_step": rows, "steps": steps, "ms_per_step": round(leg["dt"] / steps * 1e3, 3),
               "workload": "%s, rows %d.. of the generator" % (cfg, start0)}
        if with_roofline and leg["sym_ms"]:
            obj["roofline_symphony"] = roofline("group_kernel<%d>" % kind, SYMPHONY_FLOPS.get(base, 720.0), leg["sym_ms"],
                                                leg["sym_samples"], leg["sym_passes"], kind=kind)
        if with_roofline and leg["far_ms"]:
            obj["roofline_faraday"] = roofline("coop_kernel<HeyvaertsProblem<%d>>" % kind, FARADAY_FLOPS.get(base, 498.0),
                                               leg["far_ms"], leg["far_samples"], leg["far_passes"], kind=kind)
        return obj

    # ---------------------------------------------------------------- primary: eight coefficients per point, per the contract
    P = args.points
    kind, mask, _, _, _ = workload.make_batch(args.config, 1)
    nsel = bin(mask & 0xFF).count("1")
    primary = timed_leg(args.config, P, args.steps, args.warmup, TABLE if args.config == "cfg2_powerlaw_8" else 0, TABLE)
    elapsed = primary["dt"]
    # (every rank takes part: the per-rank kernel-time spread is a collective)
    base_cfg = args.config
    roof = roofline("group_kernel<%d>" % kind, SYMPHONY_FLOPS.get(base_cfg, 720.0), primary["sym_ms"],
                    primary["sym_samples"], primary["sym_passes"], with_traffic=True, kind=kind) if primary["sym_ms"] else None
    roof_far = roofline("coop_kernel<HeyvaertsProblem<%d>>" % kind, FARADAY_FLOPS.get(base_cfg, 498.0), primary["far_ms"],
                        primary["far_samples"], primary["far_passes"], with_traffic=True, kind=kind) if primary["far_ms"] else None
    tail = ctx.last_tail()


    # ---------------------------------------------------------------- side legs
    two = thermal = corner = gmin1 = pitchypl = pitchykappa = None
    if args.side_rows > 0:
        R = args.side_rows
        two = side_leg("cfg2_powerlaw_jI_aI", min(4 * R, 262144), 2, 0)
        thermal = side_leg("cfg3_thermal_8", R, 2, 0)
        # configs[3] / configs[4]: the two anisotropic tables -- the crank-out product's actual workload
        # (examples/crank-out-pitchypl.rs:157-195, crank-out-pitchykappa.rs:184-217); one warm-up + one timed step each
        pitchypl = side_leg("cfg4_pitchypl_8", R, 1, 0)
        pitchykappa = side_leg("cfg5_pitchykappa_8", R, 1, 0)
        corner = side_leg("cfg2_powerlaw_8_corner", min(R, 2048), 1, TABLE, with_roofline=False)
        gmin1 = side_leg("cfg2_powerlaw_8_gmin1", min(R, 16384), 1, TABLE, with_roofline=False)

    # ---------------------------------------------------------------- parity vs the literal-flavour vectors (rank 0)
    parity = None
    if rank == 0 and not args.no_parity:
        parity = {}
        control = {}
        for cfg in ("cfg2_powerlaw_jI_aI", "cfg2_powerlaw_8", "cfg3_thermal_8", "cfg4_pitchypl_8", "cfg5_pitchykappa_8"):
            path = os.path.join(ROOT, "tests", "golden", "literal_%s.npz" % cfg)
            if not os.path.exists(path):
                continue
            z = np.load(path)
            n, start, m = int(z["n"]), int(z["start"]), int(z["mask"])
            k2, _, s2, th2, p2 = workload.make_batch(cfg, n, start=start)
            got = ctx.compute_batch(k2, s2, th2, p2, m)
            parity[cfg] = workload.compare_tables(got, z["out"], m)
            # the noise floor: the literal flavour against two equally legitimate builds of itself (no GPU involved)
            for key in ("out_rev", "out_fma"):
                if key in z.files:
                    control.setdefault(cfg, {})[key] = workload.compare_tables(z[key], z["out"], m)
        parity["control"] = control
        parity["reference"] = ("tests/golden/literal_*.npz: out = oracle/liboracle_libm.so (glibc libm, unfused, GSL summation "
                               "order); control: out_rev = the same with the GK31 terms added in reverse order, out_fma = the "
                               "same compiled with -ffp-contract=fast; made by tools/make_literal_fixtures.py [--controls]")

    if rank == 0:
        points = P * world * args.steps
        value = points / elapsed

        cpu = None
        if args.cpu_sample > 0 and world == 1:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_bind
            L = oracle_bind.load("det")
            cores = min(len(os.sched_getaffinity(0)), 16)   # a 1-GPU box's CPU share is 16 cores
            model = "unknown"
            try:
                for ln in open("/proc/cpuinfo"):
                    if ln.startswith("model name"):
                        model = ln.split(":", 1)[1].strip()
                        break
            except OSError:
                pass

            def cpu_rate(cfg, rows, threads, start):
                k, m, s, th, params = workload.make_batch(cfg, rows, start=start)
                c0 = time.perf_counter()
                oracle_bind.batch(L, k, s, th, params, m, nthreads=threads)
                return round(rows / (time.perf_counter() - c0), 4)

            n8 = args.cpu_sample
            variants = {
                "eight_coeff_all_cores": {"value": cpu_rate("cfg2_powerlaw_8", n8, cores, TABLE), "cores": cores, "rows": n8},
                "eight_coeff_one_core": {"value": cpu_rate("cfg2_powerlaw_8", max(n8 // 16, 8), 1, TABLE), "cores": 1, "rows": max(n8 // 16, 8)},
                "two_coeff_all_cores": {"value": cpu_rate("cfg2_powerlaw_jI_aI", 4 * n8, cores, 0), "cores": cores, "rows": 4 * n8},
                "two_coeff_one_core": {"value": cpu_rate("cfg2_powerlaw_jI_aI", max(n8 // 4, 16), 1, 0), "cores": 1, "rows": max(n8 // 4, 16)},
            }
            cpu = {"value": variants["eight_coeff_all_cores"]["value"], "unit": "points/s", "cores": cores, "kind": "port",
                   "cpu_model": model,
                   "sample": "first %d rows of the same table (rows %d..), all eight coefficients, oracle/liboracle.so, OpenMP "
                             "dynamic over points" % (n8, TABLE),
                   "variants": variants, "reference_hint": REFERENCE_HINT}

        line = {
            "metric": "eight-coefficient parameter-points/sec (power law: j_I, alpha_I, j_Q, alpha_Q, j_V, alpha_V, rho_Q, rho_V per point)",
            "value": round(value, 2), "unit": "points/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE metric: power_law, random (s, theta, p, gamma_min) points (rows %d.. of the 1e6-row "
                                   "generator), all 8 coefficients, fp64; step = %d rows per GPU" % (TABLE, P),
                       "points_per_step_per_gpu": P, "coefficients_per_point": nsel,
                       "coefficients_per_s": round(value * nsel, 2), "sharding": "interleaved, gather to rank 0",
                       "target_note": "BASELINE's target is 1e7 points/s on 8 GPUs; no published number exists (vs_baseline null). "
                                      "At 100 % of the fp64 vector peak the reference's per-coefficient algorithm tops out near "
                                      "4.5e4 points/s per GPU (about 1.7e9 algorithmic flops per point): the target is beyond the roofline of the faithful algorithm"},
            "roofline": roof, "roofline_faraday": roof_far,
            "tail_of_last_step": {"symphony_heaviest_coefficient_batches": tail["symphony_heaviest_batches"],
                                  "faraday_heaviest_coefficient_batches": tail["faraday_heaviest_batches"],
                                  "faraday_heaviest_row": tail["faraday_heaviest_row"],
                                  "note": "the sequential chain of batches of the heaviest task bounds how early a launch can end"},
            "cpu_baseline": cpu,
            "two_coeff": two, "thermal_eight": thermal, "pitchypl_eight": pitchypl, "pitchykappa_eight": pitchykappa,
            "corner_theta_lt_0.05": corner, "gamma_min_1": gmin1, "parity": parity,
            "shared_mode": int(ctx.shared_mode()) if hasattr(ctx, "shared_mode") else None,
        }
        print(json.dumps(line))

    ctx.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
