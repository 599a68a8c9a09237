R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_drnn -- python3 $R/tools/lab/drnn_step.py > $R/gpurun_out/prof_drnn.log 2>&1
cd $R
python tools/prof_summary.py $(ls gpurun_out/prof_drnn/*/*kernel_trace.csv | head -1) 40 > gpurun_out/r2_drnn_by_shape.txt
rm -rf gpurun_out/prof_drnn
tail -3 gpurun_out/prof_drnn.log
head -45 gpurun_out/r2_drnn_by_shape.txt
