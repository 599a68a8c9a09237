# GPU call B of a round (after call A's profiles/r03_bench_streams1_by_launch_shape.txt is in place): HBM-side traffic of
# the roofline kernel (two --pmc passes), the general2 attention line, then the default bench.py line with its CPU baseline and the MELD / configuration-5 lines.
set -e
R=$PWD
O=$R/gpurun_out
bash tools/traffic_pmc.sh
cp $O/r03_wgrad_traffic.json profiles/r03_wgrad_traffic.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_g2 -- python3 $R/tools/lab/general2_prof.py > $O/prof_g2.log 2>&1
cd $R
python3 tools/general2_line.py $(ls $O/prof_g2/*/*kernel_stats.csv | head -1) > $O/r03_general2_line.txt
rm -rf $O/prof_g2
cat $O/r03_general2_line.txt
timeout -k 10 900 python bench.py > $O/r03_bench_default.json 2> $O/r03_bench_default.err
cat $O/r03_bench_default.json
for c in meld drnn; do
  timeout -k 10 300 python bench.py --config $c --no-cpu-baseline > $O/r03_bench_$c.json 2> $O/r03_bench_$c.err
  cat $O/r03_bench_$c.json
done
