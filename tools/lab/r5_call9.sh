set -o pipefail
R=$PWD; O=$R/gpurun_out; mkdir -p $O
for i in 1 2 3; do
  python bench.py --replay-family wgrad 2>/dev/null | grep '^{' | sed 's/^/tree /' | tee -a $O/r5_c9.log
  GANFFN_LIB=$R/gan_ffn_amd/lib/libganffn_prev.so python bench.py --replay-family wgrad 2>/dev/null | grep '^{' | sed 's/^/prev /' | tee -a $O/r5_c9.log
done
bash tools/lab/lib_ab.sh gan_ffn_amd/lib/libganffn_prev.so | tee -a $O/r5_c9.log
