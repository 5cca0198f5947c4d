set -e
for i in 1 2; do
echo "== rolled"; timeout -k 10 300 python bench.py --also none --no-cpu-baseline --steps 20 --warmup 5 2>&1 >/dev/null | grep -E "ms per iteration|solve_iter|transform6" | cut -c1-130
echo "== plain"; WMF_ROLLED=0 timeout -k 10 300 python bench.py --also none --no-cpu-baseline --steps 20 --warmup 5 2>&1 >/dev/null | grep -E "ms per iteration|solve_iter|transform6" | cut -c1-130
done
