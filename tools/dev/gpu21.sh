#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/g21_ab.txt
: > $L
timeout -k 10 600 python tools/ab_libs.py rimphony_amd/librimphony_hip.so rimphony_amd/librimphony_h128.so cfg2_powerlaw_8 16384 0x3f 3 >> $L 2>&1 && \
timeout -k 10 600 python tools/ab_libs.py rimphony_amd/librimphony_hip.so rimphony_amd/librimphony_h256.so cfg2_powerlaw_8 16384 0x3f 3 >> $L 2>&1 && \
timeout -k 10 600 python tools/ab_libs.py rimphony_amd/librimphony_hip.so rimphony_amd/librimphony_h256.so cfg5_pitchykappa_8 4096 0x3f 2 >> $L 2>&1
echo "exit $?" >> $L
cat $L
