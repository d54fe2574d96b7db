#!/bin/bash
# final build of the round, part 1: GPU suite, PMC of the group and the solo kernel
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_group gpurun_out/pmc_solo
(timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > gpurun_out/g28_tests.log 2>&1
cat gpurun_out/g28_tests.log
grep -q passed gpurun_out/g28_tests.log && ! grep -q failed gpurun_out/g28_tests.log || exit 1
RIMPHONY_SYM_SOLO=0 bash tools/pmc_collect.sh gpurun_out/pmc_group 65536 cfg2_powerlaw_8 0x3f > gpurun_out/g28_pmc.log 2>&1 && \
RIMPHONY_SYM_SOLO=1 bash tools/pmc_collect.sh gpurun_out/pmc_solo 65536 cfg2_powerlaw_8 0x3f >> gpurun_out/g28_pmc.log 2>&1
echo "pmc exit $?" >> gpurun_out/g28_pmc.log; cat gpurun_out/g28_pmc.log
cp gpurun_out/pmc_group/summary.json gpurun_out/g28_pmc_group.json; cp gpurun_out/pmc_solo/summary.json gpurun_out/g28_pmc_solo.json
