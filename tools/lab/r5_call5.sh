set -o pipefail
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_bench_paths.py -x -q -m gpu > $O/r5_c5_tests.log 2>&1; rc=$?
tail -5 $O/r5_c5_tests.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR)|Error|assert" $O/r5_c5_tests.log | head -30; tail -40 $O/r5_c5_tests.log; exit $rc; }
bash tools/dist1_ab.sh 2>&1 | tail -3
