set -e
echo "== default"; timeout -k 10 300 python tools/f64_lab.py cfg3 2>&1 | grep -E "solve64it|ms_per_iteration" | cut -c1-150
for v in "$@"; do
  echo "== $v"
  WMF_HIP_LIB=$PWD/lab_libs/lib_$v.so timeout -k 10 300 python tools/f64_lab.py cfg3 2>&1 | grep -E "solve64it|ms_per_iteration|rror" | cut -c1-150
done
