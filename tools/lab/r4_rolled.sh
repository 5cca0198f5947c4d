set -e
mkdir -p gpurun_out
for i in 1 2; do
echo "== rolled"; timeout -k 10 300 python tools/kernel_lab.py cfg3 items 0 8 2>&1 | grep -E "^flags|iteration kernel" | cut -c1-200
echo "== plain split layout"; WMF_ROLLED=0 timeout -k 10 300 python tools/kernel_lab.py cfg3 items 0 8 2>&1 | grep -E "^flags|iteration kernel" | cut -c1-200
done
true
true
