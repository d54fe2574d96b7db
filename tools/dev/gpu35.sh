#!/bin/bash
# early help for long Faraday tasks: A (off) against B (on, default threshold), same bits required
mkdir -p gpurun_out
L=gpurun_out/g35_ab.txt
: > $L
timeout -k 10 900 python tools/ab_env.py "RIMPHONY_EARLY_HELP_BATCHES=0" "RIMPHONY_EARLY_HELP_BATCHES=512" cfg5_pitchykappa_8 16384 0xc0 1 >> $L 2>&1 && \
timeout -k 10 600 python tools/ab_env.py "RIMPHONY_EARLY_HELP_BATCHES=0" "RIMPHONY_EARLY_HELP_BATCHES=512" cfg3_thermal_8 65536 0xc0 2 >> $L 2>&1 && \
timeout -k 10 600 python tools/ab_env.py "RIMPHONY_EARLY_HELP_BATCHES=0" "RIMPHONY_EARLY_HELP_BATCHES=512" cfg2_powerlaw_8 65536 0xc0 2 1000000 >> $L 2>&1
echo "exit $?" >> $L
cat $L
