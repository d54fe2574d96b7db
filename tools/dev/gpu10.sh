mkdir -p gpurun_out
R=$(pwd)
(timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > gpurun_out/g10_tests.log 2>&1
cat gpurun_out/g10_tests.log
RIMPHONY_SYM_SOLO=0 bash tools/pmc_collect.sh gpurun_out/pmc_group 65536 cfg2_powerlaw_8 0x3f > gpurun_out/g10_pmc.log 2>&1 && \
RIMPHONY_SYM_SOLO=1 bash tools/pmc_collect.sh gpurun_out/pmc_solo 65536 cfg2_powerlaw_8 0x3f >> gpurun_out/g10_pmc.log 2>&1
echo "pmc exit $?" >> gpurun_out/g10_pmc.log; cat gpurun_out/g10_pmc.log
cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-parity > $R/gpurun_out/g10_bench_under_rocprof.json 2> $R/gpurun_out/g10_rocprof.err
echo "rocprof exit $?"; cd $R
ls gpurun_out/prof_bench/*/ | head
