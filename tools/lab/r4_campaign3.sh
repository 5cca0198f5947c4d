set -e
mkdir -p gpurun_out
rm -f gpurun_out/r4_campaign3.log
( timeout -k 10 900 python tests/scale/fuzz_parity.py 1500 91 ; echo "fuzz_parity rc=$?" ) >> gpurun_out/r4_campaign3.log 2>&1
( timeout -k 10 300 python tests/scale/fuzz_train.py 300 92 ; echo "fuzz_train rc=$?" ) >> gpurun_out/r4_campaign3.log 2>&1
( timeout -k 10 300 python tests/scale/fuzz_f64.py 500 93 ; echo "fuzz_f64 rc=$?" ) >> gpurun_out/r4_campaign3.log 2>&1
( timeout -k 10 300 python tests/scale/fuzz_csr.py 300 94 ; echo "fuzz_csr rc=$?" ) >> gpurun_out/r4_campaign3.log 2>&1
( timeout -k 10 300 python tests/scale/fuzz_topn.py 150 95 ; echo "fuzz_topn rc=$?" ) >> gpurun_out/r4_campaign3.log 2>&1
grep -E "rc=|worst|cases in|ABOVE|differ|identical|miss|exact" gpurun_out/r4_campaign3.log | tail -20
