# lab: kernel trace of the default (3-stream, graph-replayed) step -> tools/lab/overlap_hist.py
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/prof_ovl -- python3 $R/bench.py --no-cpu-baseline --step-only --warmup 10 --steps 20 "$@" > $O/prof_ovl.log 2>&1 || { tail -20 $O/prof_ovl.log; exit 1; }
python3 $R/tools/lab/overlap_hist.py $(ls $O/prof_ovl/*/*kernel_trace.csv | head -1) | tee $O/overlap${TAG}.txt
rm -rf $O/prof_ovl
