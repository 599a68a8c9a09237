set -o pipefail
R=$PWD; O=$R/gpurun_out; mkdir -p $O
for i in 1 2; do
  for v in "" _prev; do
    GANFFN_LIB=$R/gan_ffn_amd/lib/libganffn$v.so python bench.py --replay-family gemm_generic 2>/dev/null | grep '^{' | sed "s/^/lib$v /" | tee -a $O/r5_c14.log
    GANFFN_LIB=$R/gan_ffn_amd/lib/libganffn$v.so python bench.py --replay-family wgrad 2>/dev/null | grep '^{' | sed "s/^/lib$v /" | tee -a $O/r5_c14.log
  done
done
bash tools/lab/lib_ab.sh gan_ffn_amd/lib/libganffn_prev.so | tee -a $O/r5_c14.log
