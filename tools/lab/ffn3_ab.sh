# lab: the step with the two-kernel feed-forward forward (mode 0) against ffn3.hip's single kernel (128: T <= 4096 only;
# 4194432 = 128 | 1 << 22: every T): on 1, 2 and 3 streams, then alternating on the default streams
for s in 1 2 3; do for m in 0 4194432; do
  GANFFN_FFN_MODE=$m python bench.py --streams $s --no-cpu-baseline --step-only 2>/dev/null | python -c "import json,sys; print('streams $s mode $m', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])" || exit 1
done; done
for rep in 1 2; do for m in 0 128 4194432; do
  GANFFN_FFN_MODE=$m python bench.py --no-cpu-baseline --step-only 2>/dev/null | python -c "import json,sys; print('rep $rep mode $m', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])" || exit 1
done; done
