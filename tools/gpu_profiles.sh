# GPU call B of a round: every artefact that goes under profiles/ (round-5 names), taken on the kernel sources in the tree:
# rocprofv3 kernel traces of the step alone (single stream, default 3 streams, configuration 5), PMC traffic of the
# weight-gradient and generic-GEMM launch mixes, SQ counters of the attention kernels, the general2 GB/s line, and the
# bench lines (default with CPU baseline, MELD dims, configuration 5).  Then copy gpurun_out/r05_* into profiles/ by hand.
set -o pipefail
R=$PWD
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GANFFN_BENCH_PREROLL=0      # traces must hold exactly warm-up + timed iterations (prof_summary.py's iteration count)
prof() {   # name, bench arguments...
  n=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$n -- python3 $R/bench.py --no-cpu-baseline "$@" > $O/prof_$n.log 2>&1 || { tail -20 $O/prof_$n.log; exit 1; }
  ( cd $R && python3 tools/prof_summary.py $(ls $O/prof_$n/*/*kernel_trace.csv | head -1) 90 ${ITER:-13} > $O/r05_${n}_by_launch_shape.txt )
  cp $(ls $O/prof_$n/*/*kernel_stats.csv | head -1) $O/r05_${n}_kernel_stats.csv
  rm -rf $O/prof_$n
}
prof bench_streams1 --streams 1 --no-graph --warmup 3 --steps 10 --step-only
prof bench_default --warmup 3 --steps 10 --step-only
ITER=70 prof drnn --config drnn --steps 10        # (60 warm-up steps + 10 timed)
cd $R
cp $O/r05_bench_streams1_by_launch_shape.txt $O/r05_drnn_by_launch_shape.txt profiles/    # bench.py reads its in-step figures from profiles/
bash tools/traffic_pmc.sh wgrad > $O/traffic_wgrad.log 2>&1 || { tail -20 $O/traffic_wgrad.log; exit 1; }
bash tools/traffic_pmc.sh gemm_generic > $O/traffic_gemm.log 2>&1 || { tail -20 $O/traffic_gemm.log; exit 1; }
cp $O/r05_wgrad_traffic.json $O/r05_gemm_traffic.json profiles/
for fam in attention ffn_k100 ffn_n100; do bash tools/family_pmc.sh $fam r05 > $O/pmc_$fam.log 2>&1 || { tail -30 $O/pmc_$fam.log; exit 1; }; done
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_g2 -- python3 $R/tools/lab/general2_prof.py > $O/prof_g2.log 2>&1 )
python3 tools/general2_line.py $(ls $O/prof_g2/*/*kernel_stats.csv | head -1) > $O/r05_general2_line.txt
rm -rf $O/prof_g2
unset GANFFN_BENCH_PREROLL
timeout -k 10 600 python bench.py > $O/r05_bench_default.json 2> $O/r05_bench_default.err || { tail -20 $O/r05_bench_default.err; exit 1; }
timeout -k 10 300 python bench.py --config meld --no-cpu-baseline > $O/r05_bench_meld.json 2> $O/r05_bench_meld.err || { tail -20 $O/r05_bench_meld.err; exit 1; }
timeout -k 10 300 python bench.py --config drnn --no-cpu-baseline > $O/r05_bench_drnn.json 2> $O/r05_bench_drnn.err || { tail -20 $O/r05_bench_drnn.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05_bench_default.json") if l.startswith("{")][0])
print("ms/step", d["ms_per_step"], "roofline", d["roofline"]["family"], d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "alg", d["roofline"]["algorithmic_bytes_per_launch"])
for f in d["roofline_families"]:
    print(" ", f["family"], f.get("share_pct"), "in-step", f.get("in_step_frac"), "live", f["frac"], f.get("avg_kernel_us"))
print("worst", d["roofline_worst"]["family"], "cpu", d.get("cpu_baseline", {}).get("value"))
print(open("gpurun_out/r05_general2_line.txt").read()[-600:])
PY
# training-level statistical parity table (tests/test_hip_train_stats.py asserts on it; this is the record for profiles/)
python3 tests/golden/make_train_stats.py hip > $O/train_stats_hip.log 2>&1 || { tail -20 $O/train_stats_hip.log; exit 1; }
python3 - <<'PY' > gpurun_out/r05_train_stats.txt
import numpy as np
c, h = np.load("tests/golden/train_stats.npz"), np.load("gpurun_out/train_stats_hip.npz")
names, cpu, hip = [str(x) for x in c["names"]], c["cpu"], h["hip"]
n = cpu.shape[0]
print("training-level statistical parity, %d seeds each: stock PyTorch on the host (tests/golden/train_stats.npz) | HIP engines on the MI355X" % n)
print("%-34s %10s %8s | %10s %8s | %s" % ("metric", "cpu mean", "std", "hip mean", "std", "|diff| / standard error"))
for j, k in enumerate(names):
    se = np.sqrt(cpu[:, j].var(ddof=1) / n + hip[:, j].var(ddof=1) / n)
    print("%-34s %10.4f %8.4f | %10.4f %8.4f | %.2f" % (k, cpu[:, j].mean(), cpu[:, j].std(ddof=1), hip[:, j].mean(), hip[:, j].std(ddof=1),
                                                        abs(cpu[:, j].mean() - hip[:, j].mean()) / max(se, 1e-12)))
PY
tail -12 gpurun_out/r05_train_stats.txt
