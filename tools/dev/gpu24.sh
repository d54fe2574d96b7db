#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/g24_ab.txt
: > $L
timeout -k 10 600 python tools/ab_env.py "RIMPHONY_FARADAY_GROUP=0" "RIMPHONY_FARADAY_GROUP=1" cfg2_powerlaw_8 16384 0xc0 2 >> $L 2>&1 && \
timeout -k 10 600 python tools/ab_env.py "RIMPHONY_FARADAY_GROUP=0" "RIMPHONY_FARADAY_GROUP=1" cfg3_thermal_8 16384 0xc0 2 >> $L 2>&1 && \
timeout -k 10 600 python tools/ab_env.py "RIMPHONY_SYM_SOLO=1" "RIMPHONY_SYM_SOLO=0" cfg2_powerlaw_8 65536 0x3f 2 >> $L 2>&1 && \
timeout -k 10 600 python tools/ab_env.py "RIMPHONY_SYM_SOLO=1" "RIMPHONY_SYM_SOLO=0" cfg2_powerlaw_8 16384 0x3f 2 >> $L 2>&1 && \
timeout -k 10 600 python tools/ab_env.py "RIMPHONY_SYM_SOLO=1" "RIMPHONY_SYM_SOLO=0" cfg3_thermal_8 16384 0x3f 2 >> $L 2>&1 && \
timeout -k 10 600 python tools/ab_env.py "RIMPHONY_SYM_SOLO=1" "RIMPHONY_SYM_SOLO=0" cfg5_pitchykappa_8 4096 0x3f 2 >> $L 2>&1 && \
timeout -k 10 600 python tools/ab_env.py "RIMPHONY_SYM_SOLO=1" "RIMPHONY_SYM_SOLO=0" cfg2_powerlaw_8 16384 0x0f 2 >> $L 2>&1 && \
timeout -k 10 600 python tools/ab_env.py "RIMPHONY_SYM_SOLO=1" "RIMPHONY_SYM_SOLO=0" cfg2_powerlaw_8 16384 0x30 2 >> $L 2>&1 && \
timeout -k 10 600 python tools/ab_env.py "RIMPHONY_SYM_SOLO=1" "RIMPHONY_SYM_SOLO=0" cfg2_powerlaw_jI_aI 65536 0x03 2 >> $L 2>&1
echo "exit $?" >> $L
cat $L
