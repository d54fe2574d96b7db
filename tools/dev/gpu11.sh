mkdir -p gpurun_out
(
timeout -k 10 500 python tools/ab_multi.py cfg2_powerlaw_8 65536 0xC0 2 1000000 variants/prev.so - &&
timeout -k 10 300 python tools/ab_multi.py cfg2_powerlaw_8 65536 0xC0 1 1131072 variants/prev.so - &&
timeout -k 10 300 python tools/ab_multi.py cfg3_thermal_8 65536 0xC0 1 0 variants/prev.so - &&
timeout -k 10 500 python tools/ab_multi.py cfg5_pitchykappa_8 16384 0xC0 1 0 variants/prev.so - &&
timeout -k 10 500 python tools/ab_multi.py cfg4_pitchypl_8 16384 0xC0 1 0 variants/prev.so -
) > gpurun_out/g11.log 2>&1
echo "exit $?" >> gpurun_out/g11.log
cat gpurun_out/g11.log
