# 1-rank RCCL path, 3 streams, with more ROCclr hardware queues (GPU_MAX_HW_QUEUES): does RCCL's internal stream stop sharing a
# hardware queue with a sub-step stream?
export MASTER_ADDR=127.0.0.1 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0
n=0
for q in 4 8 16; do
  for mode in plain one per; do
    n=$((n+1))
    if [ $mode = plain ]; then e=""; elif [ $mode = one ]; then e="GANFFN_FORCE_DIST=1"; else e="GANFFN_FORCE_DIST=1 GANFFN_COMM_PER_STREAM=1"; fi
    env GPU_MAX_HW_QUEUES=$q MASTER_PORT=$((29570+n)) $e python bench.py --no-cpu-baseline --step-only 2>&1 | grep "ms/step" | sed "s/^/hwq=$q $mode: /"
  done
done
