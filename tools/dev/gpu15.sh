#!/bin/bash
# region timers and hit counters of the group kernel
export RIMPHONY_HIP_LIB=rimphony_amd/librimphony_prof.so
mkdir -p gpurun_out
timeout -k 10 300 python tools/region_profile_group.py cfg2_powerlaw_8 16384 0x3F > gpurun_out/g15_regions.txt 2>&1 && \
timeout -k 10 300 python tools/region_profile_group.py cfg2_powerlaw_8 16384 0x0F >> gpurun_out/g15_regions.txt 2>&1 && \
timeout -k 10 300 python tools/region_profile_group.py cfg2_powerlaw_8 16384 0x30 >> gpurun_out/g15_regions.txt 2>&1 && \
timeout -k 10 300 python tools/region_profile_group.py cfg3_thermal_8 16384 0x3F >> gpurun_out/g15_regions.txt 2>&1
echo "exit $?"
cat gpurun_out/g15_regions.txt
