set -o pipefail
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_lstm.py tests/test_hip_dialogue_rnn.py -x -q -m gpu > $O/r5_c6_tests.log 2>&1; rc=$?
tail -25 $O/r5_c6_tests.log
exit $rc
