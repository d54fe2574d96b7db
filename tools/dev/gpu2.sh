mkdir -p gpurun_out
(
timeout -k 10 500 python tools/ab_env.py RIMPHONY_SYM_SOLO=1 RIMPHONY_SYM_SOLO=0 cfg2_powerlaw_8 16384 0x3f 2 1000000 &&
timeout -k 10 300 python tools/ab_env.py RIMPHONY_SYM_SOLO=1 RIMPHONY_SYM_SOLO=0 cfg2_powerlaw_jI_aI 32768 0x3 2 &&
timeout -k 10 300 python tools/ab_env.py RIMPHONY_SYM_SOLO=1 RIMPHONY_SYM_SOLO=0 cfg3_thermal_8 16384 0x3f 1 &&
timeout -k 10 300 python tools/ab_env.py RIMPHONY_SYM_SOLO=1 RIMPHONY_SYM_SOLO=0 cfg5_pitchykappa_8 8192 0x3f 1
) > gpurun_out/g2_ab.log 2>&1
echo "exit $?" >> gpurun_out/g2_ab.log
cat gpurun_out/g2_ab.log
