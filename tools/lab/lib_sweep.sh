# lab: the step on the library in the tree and on other builds of it (GANFFN_LIB), two interleaved passes
# usage: bash tools/lab/lib_sweep.sh gan_ffn_amd/lib/libganffn_a.so gan_ffn_amd/lib/libganffn_b.so ...
for rep in 1 2; do
  python bench.py --no-cpu-baseline --step-only 2>/dev/null | python -c "import json,sys; print('rep $rep tree', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])" || exit 1
  for l in "$@"; do
    GANFFN_LIB=$PWD/$l python bench.py --no-cpu-baseline --step-only 2>/dev/null | python -c "import json,sys; print('rep $rep $l', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])" || exit 1
  done
done
