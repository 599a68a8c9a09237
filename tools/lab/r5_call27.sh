set -o pipefail
R=$PWD
for rep in 1 2 3; do
  python bench.py --no-cpu-baseline --step-only 2>/dev/null | python -c "import json,sys; print('rep $rep tree', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
  GANFFN_LIB=$R/gan_ffn_amd/lib/libganffn_w3.so python bench.py --no-cpu-baseline --step-only 2>/dev/null | python -c "import json,sys; print('rep $rep wres-3-per-CU', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
done
python bench.py --replay-family ffn_k100 2>/dev/null | grep '^{' | sed 's/^/tree /'
GANFFN_LIB=$R/gan_ffn_amd/lib/libganffn_w3.so python bench.py --replay-family ffn_k100 2>/dev/null | grep '^{' | sed 's/^/w3 /'
