set -o pipefail
R=$PWD; O=$R/gpurun_out; mkdir -p $O
for i in 1 2; do
  for m in 0 1024 1536 768; do
    GANFFN_FFN_MODE=$m python bench.py --steps 60 --warmup 30 --no-cpu-baseline --step-only 2>/dev/null | grep '^{' | python -c "import json,sys; print('n100 chunks mode $m', json.loads(sys.stdin.read())['ms_per_step'])" | tee -a $O/r5_c21.log
  done
done
