set -o pipefail
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export GANFFN_ADAM_PARTS=1
GANFFN_FFN_MODE=67108864 bash tools/prof_one.sh r5_newrc --streams 1 --no-graph --warmup 3 --steps 10 --step-only || exit 1
GANFFN_FFN_MODE=201326592 bash tools/prof_one.sh r5_oldrc --streams 1 --no-graph --warmup 3 --steps 10 --step-only || exit 1
grep -E "rc_|total GPU|adam|reduce|Fill" $O/r5_newrc_by_launch_shape.txt | head -40
echo ----
grep -E "rc_|total GPU|adam|reduce|Fill" $O/r5_oldrc_by_launch_shape.txt | head -40
