R=$PWD; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export GANFFN_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dist1 -- python3 $R/bench.py --no-cpu-baseline --streams 1 --no-graph --warmup 3 --steps 10 --step-only > $O/prof_dist1.log 2>&1 || tail -20 $O/prof_dist1.log
cd $R && python3 tools/prof_summary.py $(ls $O/prof_dist1/*/*kernel_trace.csv | head -1) 60 13 > $O/r04_dist1_streams1_by_launch_shape.txt
rm -rf $O/prof_dist1
grep "ms/step" $O/prof_dist1.log
head -45 $O/r04_dist1_streams1_by_launch_shape.txt
grep "total GPU" $O/r04_dist1_streams1_by_launch_shape.txt
for s in 1 3; do python bench.py --no-cpu-baseline --streams $s --no-graph --step-only 2>&1 | grep "ms/step"; done
unset GANFFN_FORCE_DIST
python bench.py --no-cpu-baseline --streams 1 --no-graph --step-only 2>&1 | grep "ms/step"
