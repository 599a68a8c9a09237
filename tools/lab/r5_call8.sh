set -o pipefail
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_ops.py -x -q -k "tn_grouped or tn100" > $O/r5_c8_tests.log 2>&1 || { tail -30 $O/r5_c8_tests.log; exit 1; }
tail -2 $O/r5_c8_tests.log
timeout -k 10 600 python -m pytest tests/test_hip_engine.py tests/test_hip_modules.py -x -q -k "reproducible or sequential or unreduced or older_launch" > $O/r5_c8_tests2.log 2>&1 || { tail -30 $O/r5_c8_tests2.log; exit 1; }
tail -2 $O/r5_c8_tests2.log
python bench.py --replay-family wgrad 2>/dev/null | grep '^{' | tee $O/r5_c8_wgrad.log
for i in 1 2; do
  python bench.py --steps 60 --warmup 30 --no-cpu-baseline --step-only 2>/dev/null | grep '^{' | cut -c1-140 | tee -a $O/r5_c8_bench.log
done
