set -e
mkdir -p gpurun_out
rm -f gpurun_out/r4_campaign.log
( timeout -k 10 400 python tests/scale/fuzz_parity.py 400 41 ; echo "fuzz_parity rc=$?" ) >> gpurun_out/r4_campaign.log 2>&1
( timeout -k 10 300 python tests/scale/fuzz_train.py 120 42 ; echo "fuzz_train rc=$?" ) >> gpurun_out/r4_campaign.log 2>&1
( timeout -k 10 200 python tests/scale/soak_determinism.py cfg3 12 3 ; echo "soak cfg3 rc=$?" ) >> gpurun_out/r4_campaign.log 2>&1
( timeout -k 10 200 python tests/scale/soak_determinism.py cfg5s 8 3 ; echo "soak cfg5s rc=$?" ) >> gpurun_out/r4_campaign.log 2>&1
grep -E "rc=|worst|cases in|ABOVE|differ|identical|miss" gpurun_out/r4_campaign.log | tail -20
