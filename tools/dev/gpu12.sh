mkdir -p gpurun_out
R=$(pwd)
(timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > gpurun_out/g12_tests.log 2>&1
cat gpurun_out/g12_tests.log
(timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/g12_bench.json 2> gpurun_out/g12_bench.err; echo "bench exit $?")
RIMPHONY_SYM_SOLO=0 bash tools/pmc_collect.sh gpurun_out/pmc_group 65536 cfg2_powerlaw_8 0x3f > gpurun_out/g12_pmc.log 2>&1 && \
RIMPHONY_SYM_SOLO=1 bash tools/pmc_collect.sh gpurun_out/pmc_solo 65536 cfg2_powerlaw_8 0x3f >> gpurun_out/g12_pmc.log 2>&1
echo "pmc exit $?" >> gpurun_out/g12_pmc.log; cat gpurun_out/g12_pmc.log
