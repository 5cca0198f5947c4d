set -e
mkdir -p gpurun_out
rm -f gpurun_out/r4_lab5.log
for c in "cfg3 items" "cfg3m items"; do
  echo "== dma $c" >> gpurun_out/r4_lab5.log
  timeout -k 10 300 python tools/kernel_lab.py $c 0,268435456 5 >> gpurun_out/r4_lab5.log 2>&1 || { tail -30 gpurun_out/r4_lab5.log; exit 1; }
  echo "== no dma $c" >> gpurun_out/r4_lab5.log
  WMF_ITER_NO_DMA=1 timeout -k 10 300 python tools/kernel_lab.py $c 0 5 >> gpurun_out/r4_lab5.log 2>&1 || { tail -30 gpurun_out/r4_lab5.log; exit 1; }
done
grep -E "^==|^flags|iteration kernel|vs flags" gpurun_out/r4_lab5.log | cut -c1-220
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "degree_classes or heavy or negative or segments or golden or weight_range" > gpurun_out/r4_t5.log 2>&1 || { tail -40 gpurun_out/r4_t5.log; exit 1; }
tail -3 gpurun_out/r4_t5.log
