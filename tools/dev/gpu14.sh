mkdir -p gpurun_out
(SWEEP_START=600000 SWEEP_SCALE=2 timeout -k 10 700 python tools/parity_sweep.py) > gpurun_out/g14_parity.log 2>&1
echo "parity exit $?" >> gpurun_out/g14_parity.log; cat gpurun_out/g14_parity.log
(timeout -k 10 500 python tools/hostile_sweep.py && HOSTILE_WIDE=1 timeout -k 10 400 python tools/hostile_sweep.py) > gpurun_out/g14_hostile.log 2>&1
echo "hostile exit $?" >> gpurun_out/g14_hostile.log; tail -15 gpurun_out/g14_hostile.log
