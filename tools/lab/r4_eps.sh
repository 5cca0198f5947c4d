set -e
for e in 1.2e-7 2.4e-7 4.8e-7 1e-6; do
echo "== WMF_ITER_EPS=$e"; WMF_ITER_EPS=$e timeout -k 10 300 python tools/kernel_lab.py cfg3 items 0 8 2>&1 | grep -E "^flags|iteration kernel" | cut -c1-150
done
