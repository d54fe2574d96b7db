#!/bin/bash
mkdir -p gpurun_out
RIMPHONY_HIP_LIB=rimphony_amd/librimphony_tail.so timeout -k 10 600 python tools/dev/faraday_variance.py > gpurun_out/g37_var.txt 2>&1
echo "exit $?"; grep -v amdgpu gpurun_out/g37_var.txt
