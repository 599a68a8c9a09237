set -o pipefail
run() { lab=$1; shift; "$@" 2>/dev/null | grep '^{' | python -c "import json,sys; print('$lab', json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2; do
  run "3 streams default      " python bench.py --steps 60 --warmup 30 --no-cpu-baseline --step-only
  run "2 streams d100|visual  " env GANFFN_STREAM_MAP=0,0,0,0,0,0,0,0,1,1,1,1 GANFFN_STREAM_PRIO=0,-1 python bench.py --streams 2 --steps 60 --warmup 30 --no-cpu-baseline --step-only
  run "2 streams, equal prio  " env GANFFN_STREAM_MAP=0,0,0,0,0,0,0,0,1,1,1,1 GANFFN_STREAM_PRIO=0,0 python bench.py --streams 2 --steps 60 --warmup 30 --no-cpu-baseline --step-only
done
