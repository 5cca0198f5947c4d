set -e
for i in 1 2; do
echo "== rolled (low kernels rebuild the bias)"; timeout -k 10 300 python tools/kernel_lab.py cfg3 users 0 8 2>&1 | grep -E "^flags" | cut -c1-200
echo "== plain"; WMF_ROLLED=0 timeout -k 10 300 python tools/kernel_lab.py cfg3 users 0 8 2>&1 | grep -E "^flags" | cut -c1-200
done
timeout -k 10 900 python -m pytest tests/test_gpu_iter.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r4_t8.log 2>&1 || { tail -40 gpurun_out/r4_t8.log; exit 1; }
tail -3 gpurun_out/r4_t8.log
