mkdir -p gpurun_out
(
timeout -k 10 500 python tools/ab_multi.py cfg2_powerlaw_8 32768 0x3f 2 1000000 variants/noshare.so - &&
timeout -k 10 300 python tools/ab_multi.py cfg3_thermal_8 32768 0x3f 1 0 variants/noshare.so - &&
timeout -k 10 500 python tools/ab_multi.py cfg5_pitchykappa_8 16384 0x3f 1 0 variants/noshare.so - &&
timeout -k 10 300 python tools/ab_multi.py cfg4_pitchypl_8 16384 0x3f 1 0 variants/noshare.so -
) > gpurun_out/g9.log 2>&1
echo "exit $?" >> gpurun_out/g9.log
cat gpurun_out/g9.log
