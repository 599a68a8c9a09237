# rocprofv3 kernel trace of one bench.py configuration -> gpurun_out/<name>_by_launch_shape.txt (+ kernel_stats.csv)
# usage: bash tools/prof_one.sh <name> <bench.py arguments...>
set -o pipefail
R=$PWD
O=$R/gpurun_out
mkdir -p $O
n=$1; shift
cd /tmp && export TMPDIR=/tmp
export GANFFN_BENCH_PREROLL=0      # traces must hold exactly warm-up + timed iterations (prof_summary.py's iteration count)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$n -- python3 $R/bench.py --no-cpu-baseline "$@" > $O/prof_$n.log 2>&1 || { tail -20 $O/prof_$n.log; exit 1; }
python3 $R/tools/prof_summary.py $(ls $O/prof_$n/*/*kernel_trace.csv | head -1) 90 ${ITER:-13} > $O/${n}_by_launch_shape.txt
cp $(ls $O/prof_$n/*/*kernel_stats.csv | head -1) $O/${n}_kernel_stats.csv
rm -rf $O/prof_$n
