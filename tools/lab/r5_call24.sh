set -o pipefail
R=$PWD; O=$R/gpurun_out; mkdir -p $O
run() { python bench.py --steps 60 --warmup 30 --no-cpu-baseline --step-only 2>/dev/null | grep '^{' | python -c "import json,sys; print('$1', json.loads(sys.stdin.read())['ms_per_step'])" | tee -a $O/r5_c24.log; }
for i in 1 2; do
  run default
  GANFFN_EARLY_GEN=1 run early_gen
  GANFFN_STREAM_PRIO=0,0,0 run prio000
  GANFFN_STREAM_PRIO=-1,-1,-1 run prio_all_high
done
