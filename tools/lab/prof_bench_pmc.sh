#!/bin/bash
# usage: tools/lab/prof_bench_pmc.sh <tag> "<pmc counters>" [bench args]: rocprofv3 --pmc on bench.py, per-kernel means
set -e
tag=$1; shift
pmc=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $out -- python3 bench.py --no-cpu-baseline --also none "$@" > $out/stdout.log 2>&1 || { tail -n 30 $out/stdout.log; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        agg[row['Kernel_Name'][:48]][row['Counter_Name']].append(float(row['Counter_Value']))
    for k, d in agg.items():
        if any(x in k for x in ('transform6', 'gram6', 'solve_low_kernel<9, 1', 'solve_pair')):
            print(k)
            for c, v in sorted(d.items()):
                v = sorted(v)
                print(f"    {c:30s} max {v[-1]:.4g}  median {v[len(v)//2]:.4g}  n {len(v)}")
PY
