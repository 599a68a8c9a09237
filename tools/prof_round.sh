set -e
R=$PWD
python bench.py > gpurun_out/r01_bench_default.json 2> gpurun_out/r01_bench_default.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_def -- python3 $R/bench.py > $R/gpurun_out/prof_def.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_s1 -- python3 $R/bench.py --streams 1 --no-graph --no-cpu-baseline > $R/gpurun_out/prof_s1.log 2>&1
cd $R
for d in prof_def prof_s1; do
  python tools/prof_summary.py $(ls gpurun_out/$d/*/*kernel_trace.csv | head -1) 70 > gpurun_out/${d}_by_shape.txt
  cp $(ls gpurun_out/$d/*/*kernel_stats.csv | head -1) gpurun_out/${d}_kernel_stats.csv
  rm -rf gpurun_out/$d
done
cat gpurun_out/r01_bench_default.json
