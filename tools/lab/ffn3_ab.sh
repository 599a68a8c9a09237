# lab: the step with the two-kernel feed-forward forward (mode 0) against ffn3.hip's single kernel (128: T <= 4096 only;
# 4194432 = 128 | 1 << 22: every T), alternating, default streams
O=gpurun_out
for rep in 1 2; do for m in 0 128 4194432; do
  GANFFN_FFN_MODE=$m python bench.py --no-cpu-baseline --step-only > $O/ffn3_ab_${rep}_${m}.json 2>/dev/null || exit 1
  python -c "import json; d=json.loads(open('$O/ffn3_ab_${rep}_${m}.json').read().strip().splitlines()[-1]); print('rep $rep mode $m', d['ms_per_step'])"
done; done
