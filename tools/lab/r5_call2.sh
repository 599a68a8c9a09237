set -o pipefail
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "attention or linear1 or one_bit" > $O/r5_c2_tests.log 2>&1; rc=$?
tail -5 $O/r5_c2_tests.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR)|Error|assert" $O/r5_c2_tests.log | head -30; exit $rc; }
timeout -k 10 300 python tools/lab/attn_split_ab.py 2>&1 | tee $O/r5_attn_split_ab.txt
timeout -k 10 900 python -m pytest tests/test_hip_modules.py tests/test_hip_engine.py tests/test_hip_properties.py -x -q -m gpu > $O/r5_c2_tests2.log 2>&1; rc=$?
tail -5 $O/r5_c2_tests2.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR)|Error|assert" $O/r5_c2_tests2.log | head -30; exit $rc; }
for i in 1 2 3; do
  python bench.py --steps 60 --warmup 30 --no-cpu-baseline --step-only 2>/dev/null | grep '^{' | sed 's/^/split /' | cut -c1-200 | tee -a $O/r5_c2_bench.log
  GANFFN_FFN_MODE=67108864 python bench.py --steps 60 --warmup 30 --no-cpu-baseline --step-only 2>/dev/null | grep '^{' | sed 's/^/whole /' | cut -c1-200 | tee -a $O/r5_c2_bench.log
done
