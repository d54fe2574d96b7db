#!/bin/bash
export RIMPHONY_HIP_LIB=rimphony_amd/librimphony_prof.so
mkdir -p gpurun_out
timeout -k 10 300 python tools/region_profile_faraday.py 16384 cfg2_powerlaw_8 > gpurun_out/g16_regions_faraday.txt 2>&1 && \
timeout -k 10 300 python tools/region_profile_faraday.py 16384 cfg3_thermal_8 >> gpurun_out/g16_regions_faraday.txt 2>&1
echo "exit $?"
cat gpurun_out/g16_regions_faraday.txt
