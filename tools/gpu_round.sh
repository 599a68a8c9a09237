# one GPU call: full -m gpu suite, then bench lines (IEMOCAP + MELD), then a kernel trace of the single-stream step
set -o pipefail
R=$PWD
mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2_gputest.log 2>&1
rc=$?
tail -5 gpurun_out/r2_gputest.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/r2_gputest.log | head -30; exit $rc; }
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r2_bench_quick.json 2> gpurun_out/r2_bench_quick.err || { tail -20 gpurun_out/r2_bench_quick.err; exit 1; }
cat gpurun_out/r2_bench_quick.json
timeout -k 10 300 python bench.py --config meld --no-cpu-baseline > gpurun_out/r2_bench_meld_quick.json 2> gpurun_out/r2_bench_meld_quick.err || { tail -20 gpurun_out/r2_bench_meld_quick.err; exit 1; }
cat gpurun_out/r2_bench_meld_quick.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_s1 -- python3 $R/bench.py --streams 1 --no-graph --no-cpu-baseline --steps 10 > $R/gpurun_out/prof_s1.log 2>&1 || { tail -20 $R/gpurun_out/prof_s1.log; exit 1; }
cd $R
python tools/prof_summary.py $(ls gpurun_out/prof_s1/*/*kernel_trace.csv | head -1) 80 > gpurun_out/r2_streams1_by_shape.txt
cp $(ls gpurun_out/prof_s1/*/*kernel_stats.csv | head -1) gpurun_out/r2_streams1_kernel_stats.csv
rm -rf gpurun_out/prof_s1
head -45 gpurun_out/r2_streams1_by_shape.txt
