#!/bin/bash
# evidence run on the final build: GPU vs deterministic oracle on rows no test uses, and the hostile sweeps
mkdir -p gpurun_out
(SWEEP_START=800000 SWEEP_SCALE=3 timeout -k 10 900 python tools/parity_sweep.py) > gpurun_out/g30_parity.log 2>&1
echo "parity exit $?" >> gpurun_out/g30_parity.log; cat gpurun_out/g30_parity.log
(timeout -k 10 500 python tools/hostile_sweep.py && HOSTILE_WIDE=1 timeout -k 10 400 python tools/hostile_sweep.py) > gpurun_out/g30_hostile.log 2>&1
echo "hostile exit $?" >> gpurun_out/g30_hostile.log; tail -12 gpurun_out/g30_hostile.log
