#!/bin/bash
mkdir -p gpurun_out
export RIMPHONY_HIP_LIB=rimphony_amd/librimphony_tail.so
timeout -k 10 300 python tools/tail_times.py cfg2_powerlaw_8 131072 > gpurun_out/g31_tail.txt 2>&1 && \
timeout -k 10 300 python tools/tail_times.py cfg3_thermal_8 65536 0 >> gpurun_out/g31_tail.txt 2>&1 && \
timeout -k 10 300 python tools/tail_times.py cfg5_pitchykappa_8 16384 0 >> gpurun_out/g31_tail.txt 2>&1
echo "exit $?"; grep -v amdgpu gpurun_out/g31_tail.txt
