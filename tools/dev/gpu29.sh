#!/bin/bash
# final build of the round, part 2: the bench line, the same under rocprofv3, region timers
mkdir -p gpurun_out
R=$(pwd)
(timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/g29_bench.json 2> gpurun_out/g29_bench.err; echo "bench exit $?")
cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench29 -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-parity > $R/gpurun_out/g29_bench_under_rocprof.json 2> $R/gpurun_out/g29_rocprof.err
echo "rocprof exit $?"; cd $R
cp gpurun_out/prof_bench29/*/*_kernel_stats.csv gpurun_out/g29_kernel_stats.csv; cp gpurun_out/prof_bench29/*/*_kernel_trace.csv gpurun_out/g29_kernel_trace.csv
RIMPHONY_HIP_LIB=rimphony_amd/librimphony_prof.so timeout -k 10 300 python tools/region_profile_group.py cfg2_powerlaw_8 65536 0x3F > gpurun_out/g29_regions.txt 2>&1
python - <<'PY'
import json
for f in ("gpurun_out/g29_bench.json", "gpurun_out/g29_bench_under_rocprof.json"):
    d=json.load(open(f))
    print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline_faraday"]["kernel_ms"])
PY
