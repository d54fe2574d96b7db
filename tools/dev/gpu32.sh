#!/bin/bash
# final sources: PMC (group, solo), bench line, bench under rocprofv3
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_group gpurun_out/pmc_solo gpurun_out/prof_bench32
R=$(pwd)
RIMPHONY_SYM_SOLO=0 bash tools/pmc_collect.sh gpurun_out/pmc_group 65536 cfg2_powerlaw_8 0x3f > gpurun_out/g32_pmc.log 2>&1 && \
RIMPHONY_SYM_SOLO=1 bash tools/pmc_collect.sh gpurun_out/pmc_solo 65536 cfg2_powerlaw_8 0x3f >> gpurun_out/g32_pmc.log 2>&1
echo "pmc exit $?" >> gpurun_out/g32_pmc.log; cat gpurun_out/g32_pmc.log
cp gpurun_out/pmc_group/summary.json gpurun_out/g32_pmc_group.json; cp gpurun_out/pmc_solo/summary.json gpurun_out/g32_pmc_solo.json
cp gpurun_out/g32_pmc_group.json profiles/r3_pmc_group_powerlaw8.json
(timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/g32_bench.json 2> gpurun_out/g32_bench.err; echo "bench exit $?")
cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench32 -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-parity > $R/gpurun_out/g32_bench_under_rocprof.json 2> $R/gpurun_out/g32_rocprof.err
echo "rocprof exit $?"; cd $R
cp gpurun_out/prof_bench32/*/*_kernel_stats.csv gpurun_out/g32_kernel_stats.csv; cp gpurun_out/prof_bench32/*/*_kernel_trace.csv gpurun_out/g32_kernel_trace.csv
python - <<'PY'
import json
for f in ("gpurun_out/g32_bench.json", "gpurun_out/g32_bench_under_rocprof.json"):
    d=json.load(open(f))
    print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline_faraday"]["kernel_ms"])
PY
