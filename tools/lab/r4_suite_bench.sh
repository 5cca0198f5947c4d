set -e
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4_suite.log 2>&1 || { tail -60 gpurun_out/r4_suite.log; exit 1; }
tail -3 gpurun_out/r4_suite.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench2.json 2> gpurun_out/r4_bench2.err
grep -E "ms per iteration|solve_iter" gpurun_out/r4_bench2.err
wc -c gpurun_out/r4_bench2.json
