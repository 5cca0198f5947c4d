set -e
mkdir -p gpurun_out
rm -f gpurun_out/r4_campaign2.log
( timeout -k 10 500 python tests/scale/fuzz_f64.py 200 77 ; echo "fuzz_f64 rc=$?" ) >> gpurun_out/r4_campaign2.log 2>&1
( timeout -k 10 300 python tests/scale/fuzz_train.py 60 78 ; echo "fuzz_train rc=$?" ) >> gpurun_out/r4_campaign2.log 2>&1
grep -E "rc=|worst|cases in|ABOVE|differ|identical|miss" gpurun_out/r4_campaign2.log | tail -20
