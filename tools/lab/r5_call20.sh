set -o pipefail
R=$PWD; O=$R/gpurun_out; mkdir -p $O
for i in 1 2 3; do
  python bench.py --steps 60 --warmup 30 --no-cpu-baseline --step-only 2>/dev/null | grep '^{' | python -c "import json,sys; print('chunks3', json.loads(sys.stdin.read())['ms_per_step'])" | tee -a $O/r5_c20.log
  GANFFN_FFN_MODE=262144 python bench.py --steps 60 --warmup 30 --no-cpu-baseline --step-only 2>/dev/null | grep '^{' | python -c "import json,sys; print('chunks4', json.loads(sys.stdin.read())['ms_per_step'])" | tee -a $O/r5_c20.log
done
GANFFN_FFN_MODE=262144 python bench.py --replay-family wgrad 2>/dev/null | grep '^{' | sed 's/^/chunks4 /' | tee -a $O/r5_c20.log
python bench.py --replay-family wgrad 2>/dev/null | grep '^{' | sed 's/^/chunks3 /' | tee -a $O/r5_c20.log
