# lab: the default step against the step with ganffn_debug_set_ffn_mode($1), alternating, three pairs
# usage: bash tools/lab/mode_step_ab.sh <mode bits>      e.g. 16777216 (1 << 24: the wide out-proj unsplit)
O=gpurun_out
for rep in 1 2 3; do for m in 0 $1; do
  GANFFN_FFN_MODE=$m python bench.py --no-cpu-baseline --step-only 2>/dev/null | python -c "import json,sys; print('rep $rep mode $m', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])" || exit 1
done; done
