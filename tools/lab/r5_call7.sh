set -o pipefail
R=$PWD; O=$R/gpurun_out; mkdir -p $O
for i in 1 2; do
  python bench.py --steps 60 --warmup 30 --no-cpu-baseline --step-only 2>/dev/null | grep '^{' | sed 's/^/default /' | cut -c1-140 | tee -a $O/r5_c7_bench.log
  GANFFN_FFN_MODE=67108864 python bench.py --steps 60 --warmup 30 --no-cpu-baseline --step-only 2>/dev/null | grep '^{' | sed 's/^/stagger1 /' | cut -c1-140 | tee -a $O/r5_c7_bench.log
  GANFFN_FFN_MODE=134217728 python bench.py --steps 60 --warmup 30 --no-cpu-baseline --step-only 2>/dev/null | grep '^{' | sed 's/^/stagger2 /' | cut -c1-140 | tee -a $O/r5_c7_bench.log
done
GANFFN_FFN_MODE=67108864 bash tools/prof_one.sh r5_stagger1 --streams 1 --no-graph --warmup 3 --steps 10 --step-only || exit 1
grep -E "rc_.*\(376" $O/r5_stagger1_by_launch_shape.txt
