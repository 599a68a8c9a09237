set -o pipefail
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_ops.py -x -q > $O/r5_c18_tests.log 2>&1 || { tail -30 $O/r5_c18_tests.log; exit 1; }
tail -1 $O/r5_c18_tests.log
timeout -k 10 900 python -m pytest tests/test_hip_engine.py tests/test_hip_modules.py tests/test_hip_headline.py -x -q > $O/r5_c18_tests2.log 2>&1 || { tail -30 $O/r5_c18_tests2.log; exit 1; }
tail -1 $O/r5_c18_tests2.log
for i in 1 2; do
  for v in "" _prev; do
    GANFFN_LIB=$R/gan_ffn_amd/lib/libganffn$v.so python bench.py --replay-family ffn_n100 2>/dev/null | grep '^{' | sed "s/^/lib$v /" | tee -a $O/r5_c18.log
    GANFFN_LIB=$R/gan_ffn_amd/lib/libganffn$v.so python bench.py --replay-family wgrad 2>/dev/null | grep '^{' | sed "s/^/lib$v /" | tee -a $O/r5_c18.log
  done
done
bash tools/lab/lib_ab.sh gan_ffn_amd/lib/libganffn_prev.so | tee -a $O/r5_c18.log
