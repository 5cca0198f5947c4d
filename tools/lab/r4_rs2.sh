set -e
mkdir -p gpurun_out
rm -f gpurun_out/r4_rs2.log
export WMF_DEBUG_FLAGS=268435456
for lib in recmodel_amd/libwmf_hip.so build/variants/librs_noguard.so; do
  echo "== $lib (no iteration kernel)" >> gpurun_out/r4_rs2.log
  WMF_HIP_LIB=$lib timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "wide_rows" >> gpurun_out/r4_rs2.log 2>&1 || echo "PYTEST FAILED for $lib" >> gpurun_out/r4_rs2.log
  WMF_HIP_LIB=$lib timeout -k 10 400 python tests/scale/fuzz_parity.py 80 12 209,225,241,193,257 >> gpurun_out/r4_rs2.log 2>&1 || echo "FUZZ FAILED for $lib" >> gpurun_out/r4_rs2.log
  WMF_HIP_LIB=$lib timeout -k 10 300 bash -c "for f in 209 225 241 257; do python tools/lab/time_wide_rows.py \$f; done" >> gpurun_out/r4_rs2.log 2>&1 || echo "TIME FAILED for $lib" >> gpurun_out/r4_rs2.log
done
grep -E "^==|passed|failed|FAILED|miss|worst|Error|ms" gpurun_out/r4_rs2.log | cut -c1-200
