mkdir -p gpurun_out
R=$(pwd)
cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-parity > $R/gpurun_out/g13_bench_under_rocprof.json 2> $R/gpurun_out/g13_rocprof.err
echo "rocprof exit $?"; cd $R
ls gpurun_out/prof_bench/*/
python - <<'PY'
import json
d=json.load(open('gpurun_out/g13_bench_under_rocprof.json'))
print(d["value"], d["roofline"]["kernel_ms"], d["roofline"]["traffic"], d["roofline"]["traffic_kind"][:80])
PY
