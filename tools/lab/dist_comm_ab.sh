# 1-rank RCCL path, 3 streams: one communicator for all sub-step streams against one per stream (GANFFN_COMM_PER_STREAM=1)
export MASTER_ADDR=127.0.0.1 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0
for i in 1 2; do
  python bench.py --no-cpu-baseline --step-only 2>&1 | grep "ms/step" | sed 's/^/plain           /'
  GANFFN_FORCE_DIST=1 MASTER_PORT=2955$i python bench.py --no-cpu-baseline --step-only 2>&1 | grep "ms/step" | sed 's/^/dist1 one comm   /'
  GANFFN_FORCE_DIST=1 GANFFN_COMM_PER_STREAM=1 MASTER_PORT=2956$i python bench.py --no-cpu-baseline --step-only 2>&1 | grep "ms/step" | sed 's/^/dist1 per-stream /'
done
