set -o pipefail
R=$PWD; O=$R/gpurun_out; mkdir -p $O
GANFFN_EARLY_GEN=1 timeout -k 10 900 python -m pytest tests/test_hip_engine.py tests/test_hip_headline.py tests/test_hip_ddp_two_ranks.py tests/test_hip_bench_paths.py -x -q > $O/r5_c25_tests.log 2>&1 || { tail -30 $O/r5_c25_tests.log; exit 1; }
tail -1 $O/r5_c25_tests.log
run() { python bench.py --steps 60 --warmup 30 --no-cpu-baseline --step-only 2>/dev/null | grep '^{' | python -c "import json,sys; print('$1', json.loads(sys.stdin.read())['ms_per_step'])" | tee -a $O/r5_c25.log; }
for i in 1 2 3 4; do
  run default
  GANFFN_EARLY_GEN=1 run early_gen
done
GANFFN_EARLY_GEN=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-330
