# SQ counters of the float64 iteration kernels (cfg2 size): three passes of rocprofv3 --pmc over tools/f64_lab.py
set -e
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
i=0
for pmc in "SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAVES"; do
  i=$((i+1)); out=gpurun_out/r4_pmc64_$i; mkdir -p $out
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $out -- python3 tools/f64_lab.py cfg2 > $out/stdout.log 2>&1 || { tail -n 30 $out/stdout.log; exit 1; }
  python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        agg[row['Kernel_Name'][:40]][row['Counter_Name']].append(float(row['Counter_Value']))
    for k, d in agg.items():
        if 'solve64it' in k:
            print(k)
            for c, v in sorted(d.items()):
                print(f"    {c:28s} max-dispatch {max(v):.4g}   (dispatches {len(v)})")
PY
done
