set -o pipefail
O=gpurun_out
mkdir -p $O
timeout -k 10 300 python tools/lab/group_bound.py > $O/r3_group_bound.txt 2>&1 || { tail -20 $O/r3_group_bound.txt; exit 1; }
cat $O/r3_group_bound.txt
for P in "" "0,0,-1" "-1,-1,0" "-1,0,0"; do
  GANFFN_STREAM_PRIO="$P" timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 --warmup 5 2> $O/r3_prio.err | python -c "import sys,json; d=json.load(sys.stdin); print('prio [$P]', d['ms_per_step'])" || { tail -5 $O/r3_prio.err; exit 1; }
done
