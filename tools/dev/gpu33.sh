#!/bin/bash
export RIMPHONY_HIP_LIB=rimphony_amd/librimphony_prof.so
mkdir -p gpurun_out
RIMPHONY_FARADAY_GROUP=1 timeout -k 10 300 python tools/region_profile_group.py cfg2_powerlaw_8 16384 0xC0 > gpurun_out/g33_regions.txt 2>&1 && \
timeout -k 10 300 python tools/region_profile_faraday.py 16384 cfg2_powerlaw_8 >> gpurun_out/g33_regions.txt 2>&1
echo "exit $?"
grep -v amdgpu gpurun_out/g33_regions.txt
