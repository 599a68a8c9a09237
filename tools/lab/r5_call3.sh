set -o pipefail
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_hip_module_path.py tests/test_hip_engine.py tests/test_hip_modules.py -x -q -m gpu > $O/r5_c3_tests.log 2>&1; rc=$?
tail -5 $O/r5_c3_tests.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR)|Error|assert" $O/r5_c3_tests.log | head -30; exit $rc; }
timeout -k 10 300 python tools/lab/k100_modes.py 2>&1 | tee $O/r5_k100_modes.txt
for i in 1 2; do
  python bench.py --steps 60 --warmup 30 --no-cpu-baseline --step-only 2>/dev/null | grep '^{' | sed 's/^/all-new /' | cut -c1-160 | tee -a $O/r5_c3_bench.log
  GANFFN_FFN_MODE=67108864 python bench.py --steps 60 --warmup 30 --no-cpu-baseline --step-only 2>/dev/null | grep '^{' | sed 's/^/attn-whole /' | cut -c1-160 | tee -a $O/r5_c3_bench.log
  GANFFN_FFN_MODE=134217728 python bench.py --steps 60 --warmup 30 --no-cpu-baseline --step-only 2>/dev/null | grep '^{' | sed 's/^/rc-4waves /' | cut -c1-160 | tee -a $O/r5_c3_bench.log
  GANFFN_ADAM_PARTS=0 python bench.py --steps 60 --warmup 30 --no-cpu-baseline --step-only 2>/dev/null | grep '^{' | sed 's/^/adam-reduce /' | cut -c1-160 | tee -a $O/r5_c3_bench.log
  GANFFN_ADAM_PARTS=0 GANFFN_FFN_MODE=201326592 python bench.py --steps 60 --warmup 30 --no-cpu-baseline --step-only 2>/dev/null | grep '^{' | sed 's/^/round4-paths /' | cut -c1-160 | tee -a $O/r5_c3_bench.log
  GANFFN_FFN_MODE=67108864 python bench.py --steps 60 --warmup 30 --no-cpu-baseline --step-only 2>/dev/null | grep '^{' | sed 's/^/attn-whole /' | cut -c1-160 | tee -a $O/r5_c3_bench.log
done
