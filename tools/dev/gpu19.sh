#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python tools/dev/faraday_shape.py > gpurun_out/g19_faraday_shape.txt 2>&1
echo "exit $?"; cat gpurun_out/g19_faraday_shape.txt
