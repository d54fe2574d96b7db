mkdir -p gpurun_out
RIMPHONY_SYM_SOLO=0 bash tools/pmc_collect.sh gpurun_out/pmc_group 16384 cfg2_powerlaw_8 0x3f > gpurun_out/g3.log 2>&1 &&
RIMPHONY_SYM_SOLO=1 bash tools/pmc_collect.sh gpurun_out/pmc_solo 16384 cfg2_powerlaw_8 0x3f >> gpurun_out/g3.log 2>&1
echo "exit $?" >> gpurun_out/g3.log
cat gpurun_out/g3.log
