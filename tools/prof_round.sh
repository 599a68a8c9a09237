# GPU call B of a round (after call A's profiles/r02_bench_streams1_by_launch_shape.txt is in place): HBM-side traffic of
# the roofline kernel (two --pmc passes), then the default bench.py line with its CPU baseline.
set -e
R=$PWD
bash tools/traffic_pmc.sh
cp gpurun_out/r02_wgrad_traffic.json profiles/r02_wgrad_traffic.json
timeout -k 10 900 python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err
cat gpurun_out/r02_bench_default.json
