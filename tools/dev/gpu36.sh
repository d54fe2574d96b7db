#!/bin/bash
mkdir -p gpurun_out
export RIMPHONY_HIP_LIB=rimphony_amd/librimphony_diag.so
timeout -k 10 300 python tools/dev/chain_latency.py cfg5_pitchykappa_8 11648 0xc0 > gpurun_out/g36_chain.txt 2>&1 && \
timeout -k 10 300 python tools/dev/chain_latency.py cfg3_thermal_8 51111 0xc0 >> gpurun_out/g36_chain.txt 2>&1
echo "exit $?"; grep -v amdgpu gpurun_out/g36_chain.txt
