set -e
mkdir -p gpurun_out
rm -f gpurun_out/r4_rs.log
for lib in recmodel_amd/libwmf_hip.so build/variants/librs_noguard.so; do
  echo "== $lib" >> gpurun_out/r4_rs.log
  WMF_HIP_LIB=$lib timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "wide_rows" >> gpurun_out/r4_rs.log 2>&1 || echo "PYTEST FAILED for $lib" >> gpurun_out/r4_rs.log
  WMF_HIP_LIB=$lib timeout -k 10 400 python tests/scale/fuzz_parity.py 60 11 209,225,241 >> gpurun_out/r4_rs.log 2>&1 || echo "FUZZ FAILED for $lib" >> gpurun_out/r4_rs.log
done
grep -E "^==|passed|failed|FAILED|miss|worst|Error" gpurun_out/r4_rs.log | cut -c1-200
