set -o pipefail
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu -k "not bench and not launcher and not ddp" > $O/r5_c19_tests.log 2>&1 || { tail -30 $O/r5_c19_tests.log; exit 1; }
tail -1 $O/r5_c19_tests.log
for i in 1 2; do
  for v in "" _prev; do
    GANFFN_LIB=$R/gan_ffn_amd/lib/libganffn$v.so python bench.py --replay-family ffn_k100 2>/dev/null | grep '^{' | sed "s/^/lib$v /" | tee -a $O/r5_c19.log
    GANFFN_LIB=$R/gan_ffn_amd/lib/libganffn$v.so python bench.py --replay-family attention 2>/dev/null | grep '^{' | sed "s/^/lib$v /" | tee -a $O/r5_c19.log
  done
done
bash tools/lab/lib_ab.sh gan_ffn_amd/lib/libganffn_prev.so | tee -a $O/r5_c19.log
