mkdir -p gpurun_out
(timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/g6_bench.json 2> gpurun_out/g6_bench.err; echo "bench exit $?"; tail -c 600 gpurun_out/g6_bench.err) 2>&1
timeout -k 10 600 python tools/nan_map.py 65536 > gpurun_out/nan_map.txt 2> gpurun_out/nan_map.err; echo "nanmap exit $?"
tail -3 gpurun_out/nan_map.err
