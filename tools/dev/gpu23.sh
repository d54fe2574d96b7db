#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/g23_ab.txt
: > $L
timeout -k 10 600 python tools/ab_libs.py rimphony_amd/librimphony_w5.so rimphony_amd/librimphony_w6.so cfg2_powerlaw_8 32768 0x3f 2 >> $L 2>&1 && \
timeout -k 10 600 python tools/ab_libs.py rimphony_amd/librimphony_hip.so rimphony_amd/librimphony_w5.so cfg3_thermal_8 32768 0x3f 2 >> $L 2>&1 && \
timeout -k 10 600 python tools/ab_libs.py rimphony_amd/librimphony_hip.so rimphony_amd/librimphony_w5.so cfg4_pitchypl_8 8192 0x3f 2 >> $L 2>&1 && \
timeout -k 10 600 python tools/ab_libs.py rimphony_amd/librimphony_hip.so rimphony_amd/librimphony_w5.so cfg5_pitchykappa_8 4096 0x3f 2 >> $L 2>&1 && \
timeout -k 10 600 python tools/ab_libs.py rimphony_amd/librimphony_hip.so rimphony_amd/librimphony_w5.so cfg2_powerlaw_jI_aI 65536 0x03 2 >> $L 2>&1
echo "exit $?" >> $L
cat $L
