set -e
mkdir -p gpurun_out
rm -f gpurun_out/r4_f64.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_iter.py -x -q -m gpu -k "float64 or f64 or pool or cores2 or dtype or integer" >> gpurun_out/r4_f64.log 2>&1 || { tail -40 gpurun_out/r4_f64.log; exit 1; }
tail -2 gpurun_out/r4_f64.log
timeout -k 10 300 python tools/f64_lab.py cfg2 >> gpurun_out/r4_f64.log 2>&1 || { tail -30 gpurun_out/r4_f64.log; exit 1; }
timeout -k 10 300 python tools/f64_lab.py cfg3 >> gpurun_out/r4_f64.log 2>&1 || { tail -30 gpurun_out/r4_f64.log; exit 1; }
grep -E "ms_per_iteration|solve64|transform64|gram64" gpurun_out/r4_f64.log | cut -c1-160
