#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python tools/dev/order_quality.py cfg2_powerlaw_8 131072 1000000 > gpurun_out/g39_order.txt 2>&1
echo "exit $?"; grep -v amdgpu gpurun_out/g39_order.txt
