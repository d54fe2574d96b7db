#!/bin/bash
# A/B: uniform hints in wave_qag_group against the build before
mkdir -p gpurun_out
L=gpurun_out/g18_ab.txt
: > $L
timeout -k 10 600 python tools/ab_libs.py rimphony_amd/librimphony_noscan.so rimphony_amd/librimphony_hip.so cfg2_powerlaw_8 32768 0x3f 3 >> $L 2>&1 && \
timeout -k 10 600 python tools/ab_libs.py rimphony_amd/librimphony_noscan.so rimphony_amd/librimphony_hip.so cfg3_thermal_8 32768 0x3f 2 >> $L 2>&1 && \
timeout -k 10 600 python tools/ab_libs.py rimphony_amd/librimphony_noscan.so rimphony_amd/librimphony_hip.so cfg5_pitchykappa_8 4096 0x3f 2 >> $L 2>&1
echo "exit $?" >> $L
cat $L
