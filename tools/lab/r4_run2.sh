set -e
mkdir -p gpurun_out
rm -f gpurun_out/r4_lab2.log
for eps in 6e-8 1.2e-7 2.4e-7; do
for c in "cfg3 items" "cfg2 items" "cfg5s items"; do
  echo "== WMF_ITER_EPS=$eps $c" >> gpurun_out/r4_lab2.log
  WMF_ITER_EPS=$eps timeout -k 10 300 python tools/kernel_lab.py $c 0,268435456 5 >> gpurun_out/r4_lab2.log 2>&1 || { tail -30 gpurun_out/r4_lab2.log; exit 1; }
done
done
grep -v "amdgpu.ids" gpurun_out/r4_lab2.log
