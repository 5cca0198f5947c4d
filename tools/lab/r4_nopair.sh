set -e
for side in items users; do
for i in 1 2; do
echo "== $side default"; timeout -k 10 300 python tools/kernel_lab.py cfg3 $side 0 8 2>&1 | grep -E "^flags" | cut -c1-330
echo "== $side pair fetch from entry 0 (results wrong, timing only)"; WMF_HIP_LIB=$PWD/lab_libs/lib_nopair.so timeout -k 10 300 python tools/kernel_lab.py cfg3 $side 0 8 2>&1 | grep -E "^flags" | cut -c1-330
done; done
