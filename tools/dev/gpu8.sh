mkdir -p gpurun_out
(
for rows in 16384 65536 131072; do timeout -k 10 400 python tools/ab_multi.py cfg2_powerlaw_8 $rows 0x3f 1 1000000 - ; done
for rows in 16384 65536; do timeout -k 10 400 python tools/ab_multi.py cfg2_powerlaw_8 $rows 0xC0 1 1000000 - ; done
) > gpurun_out/g8.log 2>&1
echo "exit $?" >> gpurun_out/g8.log
cat gpurun_out/g8.log
