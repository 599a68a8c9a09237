# round 5, GPU call 1: launcher tests, FFN-pair counters, bench line with pre-roll at the driver's --steps 20 --warmup 5 vs 30 + 60
set -o pipefail
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_bench_paths.py -x -q -m gpu > $O/r5_c1_tests.log 2>&1; rc=$?
tail -5 $O/r5_c1_tests.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR)|Error|assert" $O/r5_c1_tests.log | head -30; exit $rc; }
for fam in ffn_k100 ffn_n100; do
  bash tools/family_pmc.sh $fam r05 > $O/pmc_$fam.log 2>&1 || { tail -30 $O/pmc_$fam.log; exit 1; }
done
for i in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --step-only 2>/dev/null | grep '^{' | sed 's/^/w5s20 /' | cut -c1-400 | tee -a $O/r5_c1_bench.log
  python bench.py --steps 60 --warmup 30 --no-cpu-baseline --step-only 2>/dev/null | grep '^{' | sed 's/^/w30s60 /' | cut -c1-400 | tee -a $O/r5_c1_bench.log
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r5_c1_full.json 2> $O/r5_c1_full.err || { tail -20 $O/r5_c1_full.err; exit 1; }
cut -c1-1500 $O/r5_c1_full.json
