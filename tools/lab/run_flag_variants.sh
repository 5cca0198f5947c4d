for v in base dl_noslp base dl_noslp; do
  if [ $v = base ]; then lib=recmodel_amd/libwmf_hip.so; else lib=build/variants/lib$v.so; fi
  WMF_HIP_LIB=$lib python bench.py --also none --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('$v cfg3', round(j['ms_per_step'],2), r['kernel'], round(r['avg_launch_ms'],2))"
done
for v in base; do
  if [ $v = base ]; then lib=recmodel_amd/libwmf_hip.so; else lib=build/variants/lib$v.so; fi
  WMF_HIP_LIB=$lib python bench.py --config cfg5s --also none --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('$v cfg5s', round(j['ms_per_step'],2), r['kernel'], round(r['avg_launch_ms'],2))"
done
