set -e
echo "== default"; timeout -k 10 300 python tools/kernel_lab.py cfg3 items 0 5 2>&1 | grep -E "iteration kernel|solve_iter" | cut -c1-200
echo "== same rows (every gather a cache hit)"; WMF_HIP_LIB=$PWD/lab_libs/lib_same.so timeout -k 10 300 python tools/kernel_lab.py cfg3 items 0 5 2>&1 | grep -E "iteration kernel|solve_iter" | cut -c1-200
