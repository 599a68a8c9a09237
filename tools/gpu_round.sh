# GPU call A of a round (round 5 names): full -m gpu suite, quick bench lines (IEMOCAP, MELD dims, configuration 5), then rocprofv3 kernel
# traces of the single-stream step, the default 3-stream step and the configuration-5 step.  Everything lands under
# gpurun_out/ as r05_*; the summaries are then copied into profiles/ by hand (gpurun only merges gpurun_out/ back).
set -o pipefail
R=$PWD
O=$R/gpurun_out
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/r05_gputest.log 2>&1
rc=$?
tail -5 $O/r05_gputest.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR)|Error|assert" $O/r05_gputest.log | head -30; exit $rc; }
for c in iemocap meld drnn; do
  timeout -k 10 300 python bench.py --config $c --no-cpu-baseline > $O/r05_bench_${c}_quick.json 2> $O/r05_bench_${c}_quick.err || { tail -20 $O/r05_bench_${c}_quick.err; exit 1; }
  cat $O/r05_bench_${c}_quick.json
done
cd /tmp && export TMPDIR=/tmp
export GANFFN_BENCH_PREROLL=0      # traces must hold exactly warm-up + timed iterations (prof_summary.py's iteration count)
prof() {   # name, bench arguments...
  n=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$n -- python3 $R/bench.py --no-cpu-baseline "$@" > $O/prof_$n.log 2>&1 || { tail -20 $O/prof_$n.log; exit 1; }
  python3 $R/tools/prof_summary.py $(ls $O/prof_$n/*/*kernel_trace.csv | head -1) 90 ${ITER:-13} > $O/r05_${n}_by_launch_shape.txt
  cp $(ls $O/prof_$n/*/*kernel_stats.csv | head -1) $O/r05_${n}_kernel_stats.csv
  rm -rf $O/prof_$n
}
prof bench_streams1 --streams 1 --no-graph --warmup 3 --steps 10 --step-only
prof bench_default --warmup 3 --steps 10 --step-only
ITER=70 prof drnn --config drnn --steps 10        # (60 warm-up steps + 10 timed)
head -40 $O/r05_bench_streams1_by_launch_shape.txt
