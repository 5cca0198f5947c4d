#!/bin/bash
# Profile bench.py on the GPU box (run from the repo root through gpurun):
#   pass 1: rocprofv3 --kernel-trace --stats          -> per-kernel average durations
#   pass 2: rocprofv3 --pmc FETCH_SIZE  (own run)     -> HBM read  KB per dispatch
#   pass 3: rocprofv3 --pmc WRITE_SIZE  (own run)     -> HBM write KB per dispatch
# usage: tools/profile_bench.sh <tag> [bench args...]
set -e
tag=$1; shift
export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline "$@" > $out/bench_traced.json 2> $out/trace.err || { tail -n 20 $out/trace.err; exit 1; }
cp $root/gpurun_out/bench_detail.json $out/bench_detail_traced.json      # (the full record of that run: per-kernel table)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --no-cpu-baseline "$@" > /dev/null 2> $out/fetch.err || { tail -n 20 $out/fetch.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --no-cpu-baseline "$@" > /dev/null 2> $out/write.err || { tail -n 20 $out/write.err; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
summary = {}
for f in glob.glob(out + '/trace/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open(out + '/kernel_stats.csv', 'w') as g:
        g.write(open(f).read())
    for r in rows[:12]:
        print({k: r[k] for k in list(r)[:7]})
for name, counter in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
    agg = collections.defaultdict(list)
    for f in glob.glob(out + f'/{name}/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if row['Counter_Name'] == counter:
                agg[row['Kernel_Name']].append((int(row['Dispatch_Id']), float(row['Counter_Value'])))
    for k, v in agg.items():
        v = [x for _, x in sorted(v)]                      # dispatch order: the two half steps of an iteration alternate
        summary.setdefault(k, {})[counter + '_KB_per_dispatch_mean'] = sum(v) / len(v)
        summary[k][counter + '_KB_per_dispatch_max'] = max(v)
        summary[k][counter + '_KB_per_dispatch'] = v[:64]
        summary[k]['dispatches'] = len(v)
json.dump(summary, open(out + '/traffic_raw.json', 'w'), indent=1)
for k, v in summary.items():
    if 'solve' in k or 'transform' in k or 'gram_kernel' in k:
        print(k[:70], v)
PY
