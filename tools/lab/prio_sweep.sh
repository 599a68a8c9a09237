# lab: the step under GANFFN_STREAM_PRIO variants (priorities of the three sub-step streams), two interleaved passes
for rep in 1 2; do for pr in ${PRIOS:-"0,0,-1" "0,0,0" "-1,0,0" "0,-1,0" "-1,-1,0"}; do
  GANFFN_STREAM_PRIO=$pr python bench.py --no-cpu-baseline --step-only 2>/dev/null | python -c "import json,sys; print('rep $rep prio $pr', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])" || exit 1
done; done
