set -e
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r4_suite.log 2>&1 || { tail -60 gpurun_out/r4_suite.log; exit 1; }
tail -5 gpurun_out/r4_suite.log
