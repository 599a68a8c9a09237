# lab: the step under a list of ganffn_debug_set_ffn_mode values (first should be 0 = default), two interleaved passes
# usage: bash tools/lab/mode_sweep.sh 0 1024 2048 ...
O=gpurun_out
for rep in 1 2; do for m in "$@"; do
  GANFFN_FFN_MODE=$m python bench.py --no-cpu-baseline --step-only 2>/dev/null | python -c "import json,sys; print('rep $rep mode $m', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])" || exit 1
done; done
