# lab: the step on the library in the tree against another build of it (GANFFN_LIB=<path>, e.g. the previous commit's), alternating
# usage: bash tools/lab/lib_ab.sh gan_ffn_amd/lib/libganffn_prev.so
O=gpurun_out
for rep in 1 2 3; do
  python bench.py --no-cpu-baseline --step-only 2>/dev/null | python -c "import json,sys; print('rep $rep tree', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])" || exit 1
  GANFFN_LIB=$PWD/$1 python bench.py --no-cpu-baseline --step-only 2>/dev/null | python -c "import json,sys; print('rep $rep other', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])" || exit 1
done
