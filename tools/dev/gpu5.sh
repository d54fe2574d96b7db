mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/g5_tests.log 2>&1
echo "exit $?" >> gpurun_out/g5_tests.log
tail -15 gpurun_out/g5_tests.log
