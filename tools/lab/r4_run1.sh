set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "degree_classes or heavy or negative or segments or golden or weight_range" > gpurun_out/r4_t1.log 2>&1 || { tail -40 gpurun_out/r4_t1.log; exit 1; }
tail -3 gpurun_out/r4_t1.log
for c in "cfg3 items" "cfg3 users" "cfg2 items" "cfg5s items"; do
  LAB_HIST=1 timeout -k 10 300 python tools/kernel_lab.py $c 0,268435456 5 >> gpurun_out/r4_lab1.log 2>&1 || { tail -30 gpurun_out/r4_lab1.log; exit 1; }
done
cat gpurun_out/r4_lab1.log
