set -e
WMF_HIP_LIB=$PWD/lab_libs/lib_ring8.so timeout -k 10 600 python -m pytest tests/test_gpu_iter.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2 3; do
echo "== ring 8"; WMF_HIP_LIB=$PWD/lab_libs/lib_ring8.so timeout -k 10 300 python bench.py --also none --no-cpu-baseline --steps 20 --warmup 5 2>&1 >/dev/null | grep -E "ms per iteration|solve_iter" | cut -c1-130
echo "== no ring"; timeout -k 10 300 python bench.py --also none --no-cpu-baseline --steps 20 --warmup 5 2>&1 >/dev/null | grep -E "ms per iteration|solve_iter" | cut -c1-130
done
