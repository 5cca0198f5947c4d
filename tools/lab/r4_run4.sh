set -e
mkdir -p gpurun_out
rm -f gpurun_out/r4_lab4.log
for lib in build/variants/libit_same.so; do
for c in "cfg3 items" "cfg2 items" "cfg5s items"; do
  echo "== $lib $c" >> gpurun_out/r4_lab4.log
  WMF_HIP_LIB=$lib timeout -k 10 300 python tools/kernel_lab.py $c 0 5 >> gpurun_out/r4_lab4.log 2>&1 || { tail -30 gpurun_out/r4_lab4.log; exit 1; }
done
done
grep -E "^==|^flags|iteration kernel" gpurun_out/r4_lab4.log | cut -c1-200
