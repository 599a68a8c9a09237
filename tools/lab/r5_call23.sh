set -o pipefail
R=$PWD; O=$R/gpurun_out; mkdir -p $O
GANFFN_LIB=$R/gan_ffn_amd/lib/libganffn_ilp.so timeout -k 10 900 python -m pytest tests/test_hip_ops.py tests/test_hip_engine.py tests/test_hip_modules.py -x -q > $O/r5_c23_tests.log 2>&1 || { tail -30 $O/r5_c23_tests.log; exit 1; }
tail -1 $O/r5_c23_tests.log
for rep in 1 2 3; do
  python bench.py --no-cpu-baseline --step-only 2>/dev/null | python -c "import json,sys; print('rep $rep tree', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
  GANFFN_LIB=$R/gan_ffn_amd/lib/libganffn_ilp.so python bench.py --no-cpu-baseline --step-only 2>/dev/null | python -c "import json,sys; print('rep $rep max-ilp', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
done
for f in attention gemm_generic ffn_k100; do
  python bench.py --replay-family $f 2>/dev/null | grep '^{' | sed 's/^/tree /'
  GANFFN_LIB=$R/gan_ffn_amd/lib/libganffn_ilp.so python bench.py --replay-family $f 2>/dev/null | grep '^{' | sed 's/^/max-ilp /'
done
