# Is "more concurrency makes the step slower" a property of the chip or of ROCclr's 4 hardware queues per process
# (GPU_MAX_HW_QUEUES, default 4: HIP streams beyond that share a queue and serialise)?  Step time with the default and with 8
# queues, for the default 3-stream step, the early-generator-forward variant (6 streams) and configuration 5 on 1 / 3 streams.
set -o pipefail
O=gpurun_out
mkdir -p $O
run() {  # label, env..., -- args
  label=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 200 python bench.py --no-cpu-baseline --step-only "$@" 2>> $O/r3_hwq.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$label', d['ms_per_step'])" || { tail -5 $O/r3_hwq.err; exit 1; }
}
drnn() {
  label=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 200 python bench.py --no-cpu-baseline --config drnn 2>> $O/r3_hwq.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$label', d['ms_per_step'])" || { tail -5 $O/r3_hwq.err; exit 1; }
}
run "default queues, 3 streams        " A=1 -- --steps 30 --warmup 5
run "8 queues, 3 streams              " GPU_MAX_HW_QUEUES=8 -- --steps 30 --warmup 5
run "default queues, early generator  " GANFFN_EARLY_GEN=1 -- --steps 30 --warmup 5
run "8 queues, early generator        " GPU_MAX_HW_QUEUES=8 GANFFN_EARLY_GEN=1 -- --steps 30 --warmup 5
run "16 queues, early generator       " GPU_MAX_HW_QUEUES=16 GANFFN_EARLY_GEN=1 -- --steps 30 --warmup 5
run "default queues, 3 streams (again)" A=1 -- --steps 30 --warmup 5
drnn "drnn default queues, 1 stream    " A=1 --
drnn "drnn default queues, 3 streams   " GANFFN_DRNN_STREAMS=3 --
drnn "drnn 8 queues, 3 streams         " GPU_MAX_HW_QUEUES=8 GANFFN_DRNN_STREAMS=3 --
drnn "drnn 8 queues, 1 stream          " GPU_MAX_HW_QUEUES=8 --
