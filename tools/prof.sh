#!/bin/bash
# usage: tools/prof.sh <tag> "<pmc counters>" <kernel_lab args...>
# Runs rocprofv3 --pmc on tools/kernel_lab.py from the repo root and leaves CSVs in gpurun_out/<tag>/
set -e
tag=$1; shift
pmc=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $out -- python3 tools/kernel_lab.py "$@" > $out/stdout.log 2>&1 || { tail -n 30 $out/stdout.log; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'][:60]
        agg[k][row['Counter_Name']] += float(row['Counter_Value'])
        n[(k, row['Counter_Name'])] += 1
    for k, d in agg.items():
        if 'solve' in k or 'transform' in k or 'gram' in k or 'factorize' in k:
            print(k)
            for c, v in sorted(d.items()):
                print(f"    {c:32s} per-dispatch {v / n[(k, c)]:.4g}   (dispatches {n[(k, c)]})")
PY
