#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/g34_ab.txt
: > $L
timeout -k 10 600 python tools/ab_libs.py rimphony_amd/librimphony_hip.so rimphony_amd/librimphony_hw6.so cfg2_powerlaw_8 32768 0xc0 2 >> $L 2>&1 && \
timeout -k 10 600 python tools/ab_libs.py rimphony_amd/librimphony_hip.so rimphony_amd/librimphony_hw6.so cfg3_thermal_8 32768 0xc0 2 >> $L 2>&1
echo "exit $?" >> $L
cat $L
