#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/g38_ab.txt
: > $L
timeout -k 10 600 python tools/ab_libs.py rimphony_amd/librimphony_b16.so rimphony_amd/librimphony_b4.so cfg2_powerlaw_8 16384 0x3f 3 >> $L 2>&1 && \
timeout -k 10 600 python tools/ab_libs.py rimphony_amd/librimphony_b16.so rimphony_amd/librimphony_b2.so cfg2_powerlaw_8 16384 0x3f 3 >> $L 2>&1 && \
timeout -k 10 600 python tools/ab_libs.py rimphony_amd/librimphony_b16.so rimphony_amd/librimphony_b4.so cfg2_powerlaw_8 65536 0x3f 2 >> $L 2>&1
echo "exit $?" >> $L
cat $L
