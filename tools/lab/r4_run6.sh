set -e
mkdir -p gpurun_out
rm -f gpurun_out/r4_lab6.log
for lib in recmodel_amd/libwmf_hip.so build/variants/libit_nw2.so; do
for c in "cfg4 items" "cfg3 items"; do
  echo "== $lib $c" >> gpurun_out/r4_lab6.log
  WMF_HIP_LIB=$lib timeout -k 10 300 python tools/kernel_lab.py $c 0,268435456 5 >> gpurun_out/r4_lab6.log 2>&1 || { tail -30 gpurun_out/r4_lab6.log; exit 1; }
done
done
grep -E "^==|^flags=0|iteration kernel|vs flags" gpurun_out/r4_lab6.log | cut -c1-200
