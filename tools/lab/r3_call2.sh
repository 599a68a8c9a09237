# lab call: parity of the touched paths, A/B of debug bit 6 (PE + layer-0 in-proj as one launch), attention backward
# timing (attn_big.py), then the step
set -o pipefail
O=gpurun_out
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_modules.py tests/test_hip_ops.py tests/test_hip_engine.py tests/test_hip_meld.py tests/test_hip_drnn_engine.py -x -q -m gpu > $O/r3_call2_tests.log 2>&1 || { tail -30 $O/r3_call2_tests.log; exit 1; }
tail -2 $O/r3_call2_tests.log
timeout -k 10 300 python tools/lab/mode_ab.py 0 64 > $O/r3_mode_ab_pe.txt 2>&1 || { tail -20 $O/r3_mode_ab_pe.txt; exit 1; }
cat $O/r3_mode_ab_pe.txt
timeout -k 10 200 python tools/lab/attn_time.py > $O/r3_attn_time.txt 2>&1 || { tail -20 $O/r3_attn_time.txt; exit 1; }
cat $O/r3_attn_time.txt
for i in 1 2; do
timeout -k 10 200 python bench.py --no-cpu-baseline --step-only --steps 30 --warmup 5 2> $O/r3_call2.err || { tail -5 $O/r3_call2.err; exit 1; }
done
