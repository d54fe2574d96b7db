#!/bin/bash
export RIMPHONY_HIP_LIB=rimphony_amd/librimphony_prof.so
mkdir -p gpurun_out
timeout -k 10 300 python tools/region_profile_group.py cfg2_powerlaw_8 16384 0x3F > gpurun_out/g20_regions.txt 2>&1 && \
timeout -k 10 300 python tools/region_profile_group.py cfg2_powerlaw_8 65536 0x3F >> gpurun_out/g20_regions.txt 2>&1
echo "exit $?"
grep -v "^  " gpurun_out/g20_regions.txt
