#!/bin/bash
# Faraday tasks of the power-law families small-s-first behind the first bucket: A (off) against B (on)
mkdir -p gpurun_out
L=gpurun_out/g40_ab.txt
: > $L
timeout -k 10 600 python tools/ab_env.py "RIMPHONY_FARADAY_ORDER=0" "RIMPHONY_FARADAY_ORDER=1" cfg2_powerlaw_8 131072 0xc0 2 1000000 >> $L 2>&1 && \
timeout -k 10 600 python tools/ab_env.py "RIMPHONY_FARADAY_ORDER=0" "RIMPHONY_FARADAY_ORDER=1" cfg2_powerlaw_8 131072 0xc0 1 1145728 >> $L 2>&1 && \
timeout -k 10 600 python tools/ab_env.py "RIMPHONY_FARADAY_ORDER=0" "RIMPHONY_FARADAY_ORDER=1" cfg2_powerlaw_8 16384 0xc0 2 1000000 >> $L 2>&1 && \
timeout -k 10 600 python tools/ab_env.py "RIMPHONY_FARADAY_ORDER=0" "RIMPHONY_FARADAY_ORDER=1" cfg4_pitchypl_8 32768 0xc0 2 >> $L 2>&1
echo "exit $?" >> $L
cat $L
RIMPHONY_FARADAY_ORDER=1 RIMPHONY_HIP_LIB=rimphony_amd/librimphony_tail.so timeout -k 10 300 python tools/tail_times.py cfg2_powerlaw_8 131072 2>&1 | grep "faraday  " 
