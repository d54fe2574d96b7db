#!/bin/bash
# A/B: vector scan of picks / pass preparation against member-by-member; md5 against the solo kernel
mkdir -p gpurun_out
L=gpurun_out/g17_ab.txt
: > $L
timeout -k 10 600 python tools/ab_env.py "RIMPHONY_SYM_SOLO=1" "RIMPHONY_SYM_SOLO=0" cfg2_powerlaw_8 8192 0x3f 1 1000000 >> $L 2>&1 && \
timeout -k 10 600 python tools/ab_env.py "RIMPHONY_SYM_SOLO=1" "RIMPHONY_SYM_SOLO=0" cfg5_pitchykappa_8 2048 0x3f 1 >> $L 2>&1 && \
timeout -k 10 600 python tools/ab_libs.py rimphony_amd/librimphony_noscan.so rimphony_amd/librimphony_hip.so cfg2_powerlaw_8 32768 0x3f 3 >> $L 2>&1 && \
timeout -k 10 600 python tools/ab_libs.py rimphony_amd/librimphony_noscan.so rimphony_amd/librimphony_hip.so cfg3_thermal_8 32768 0x3f 2 >> $L 2>&1
echo "exit $?" >> $L
cat $L
