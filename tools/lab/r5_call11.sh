set -o pipefail
R=$PWD; O=$R/gpurun_out; mkdir -p $O
for i in 1 2; do
  for v in _abl2 _abl3; do
    GANFFN_LIB=$R/gan_ffn_amd/lib/libganffn$v.so python bench.py --replay-family wgrad 2>/dev/null | grep '^{' | sed "s/^/lib$v /" | tee -a $O/r5_c11.log
  done
done
