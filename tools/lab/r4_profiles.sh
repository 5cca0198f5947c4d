set -e
for cfg in cfg3 cfg2 cfg5s; do
  bash tools/profile_bench.sh prof_r04_$cfg --config $cfg --also none --steps 5 --warmup 1 > gpurun_out/prof_r04_$cfg.log 2>&1 || { tail -30 gpurun_out/prof_r04_$cfg.log; exit 1; }
  tail -4 gpurun_out/prof_r04_$cfg.log | cut -c1-200
done
