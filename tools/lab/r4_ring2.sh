set -e
for i in 1 2 3 4; do
echo "== ring 8"; timeout -k 10 300 python tools/kernel_lab.py cfg3 items 0 12 2>&1 | grep -E "^flags" | cut -c1-80
echo "== no ring"; WMF_HIP_LIB=$PWD/lab_libs/lib_ring0.so timeout -k 10 300 python tools/kernel_lab.py cfg3 items 0 12 2>&1 | grep -E "^flags" | cut -c1-80
done
