# The N = 1 point of the scaling curve on one GPU: the plain engine against the data-parallel code path with a 1-rank RCCL
# group (GANFFN_FORCE_DIST=1: bucketed backward in 4-5 layer ranges, async all-reduce per bucket, Adam per bucket), same
# box, interleaved, step only.  Writes gpurun_out/r05_bench_dist1.json.
R=$PWD
O=$R/gpurun_out
mkdir -p $O
: > $O/dist1_ab.log
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --step-only 2>/dev/null | grep '^{' | sed 's/^/plain /' >> $O/dist1_ab.log
  GANFFN_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2953$i WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 python bench.py --no-cpu-baseline --step-only 2>$O/dist1_err_$i.log | grep '^{' | sed 's/^/dist1 /' >> $O/dist1_ab.log
done
cat $O/dist1_ab.log | cut -c1-220
python3 - <<'PY'
import json
rows = {"plain": [], "dist1": []}
for l in open("gpurun_out/dist1_ab.log"):
    k, j = l.split(" ", 1)
    rows[k].append(json.loads(j)["ms_per_step"])
p, d = min(rows["plain"]), min(rows["dist1"])
out = {"what": "ms per GAN step at 1 rank, 3 streams, step only: the plain engine against the data-parallel path with a 1-rank RCCL group "
               "(GANFFN_FORCE_DIST=1), same box, interleaved runs", "plain_ms": rows["plain"], "dist1_ms": rows["dist1"],
       "best_plain_ms": p, "best_dist1_ms": d, "overhead_pct": round(100 * (d / p - 1), 2)}
json.dump(out, open("gpurun_out/r05_bench_dist1.json", "w"), indent=1)
print(json.dumps(out))
PY
