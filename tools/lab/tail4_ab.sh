# lab: the step with rows 96..99 of the 100-wide products on v_mfma_f32_4x4x1 (mode 0, default) against the padded seventh
# tile (8388608 = 1 << 23), alternating, default streams
O=gpurun_out
for rep in 1 2 3; do for m in 0 8388608; do
  GANFFN_FFN_MODE=$m python bench.py --no-cpu-baseline --step-only > $O/tail4_ab_${rep}_${m}.json 2>/dev/null || exit 1
  python -c "import json; d=json.loads(open('$O/tail4_ab_${rep}_${m}.json').read().strip().splitlines()[-1]); print('rep $rep mode $m', d['ms_per_step'])"
done; done
