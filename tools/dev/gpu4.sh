mkdir -p gpurun_out
(
timeout -k 10 900 python tools/ab_multi.py cfg2_powerlaw_8 16384 0x3f 2 1000000 -,RIMPHONY_SYM_SOLO=1 - variants/w3.so variants/w5.so variants/w4c32.so &&
timeout -k 10 300 python tools/ab_multi.py cfg2_powerlaw_8 16384 0x0f 1 1000000 -,RIMPHONY_SYM_SOLO=1 - &&
timeout -k 10 300 python tools/ab_multi.py cfg2_powerlaw_8 16384 0x30 1 1000000 -,RIMPHONY_SYM_SOLO=1 - &&
timeout -k 10 300 python tools/ab_multi.py cfg2_powerlaw_jI_aI 32768 0x3 1 0 -,RIMPHONY_SYM_SOLO=1 - variants/w5.so
) > gpurun_out/g4_ab.log 2>&1
echo "exit $?" >> gpurun_out/g4_ab.log
cat gpurun_out/g4_ab.log
