# final GPU call of the round: full -m gpu suite, then every profile artefact on the final kernel sources
set -o pipefail
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/r05_gputest.log 2>&1; rc=$?
tail -4 $O/r05_gputest.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR)|Error|assert" $O/r05_gputest.log | head -30; exit $rc; }
bash tools/gpu_profiles.sh 2>&1 | tail -25
