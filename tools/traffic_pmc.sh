#!/bin/bash
# HBM-side traffic of the roofline kernel (gemm_tn_grouped_kernel), per MI355X_MICROARCH.md "HBM": FETCH_SIZE and
# WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (kernel trace only), FETCH_SIZE doubled on gfx950.  Run from the repo
# root on the GPU box; writes gpurun_out/r03_wgrad_traffic.json (copied into profiles/ afterwards) (+ the two raw per-dispatch CSVs under gpurun_out/).
set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --replay-dominant-only > $R/gpurun_out/pmc_$c.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, json
def per_launch(counter):
    # one LOGICAL weight-gradient launch = tn100_kernel + tn100_reduce_kernel (d_model 100) or gemm_tn_grouped_kernel (512)
    f = glob.glob("gpurun_out/pmc_%s/*/*counter_collection.csv" % counter)[0]
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        if "tn100_kernel" in k or "gemm_tn_grouped_kernel" in k:
            tot += float(r["Counter_Value"]); n += 1
        elif "tn100_reduce_kernel" in k:
            tot += float(r["Counter_Value"])
    return tot / n, n
fetch_kb, n1 = per_launch("FETCH_SIZE")
write_kb, n2 = per_launch("WRITE_SIZE")
out = {"kernel": "grouped weight-gradient launch (tn100_kernel + tn100_reduce_kernel | gemm_tn_grouped_kernel)", "seq_len": 94, "dialogues_per_gpu": 32, "dispatches_profiled": [n1, n2],
       "FETCH_SIZE_KB_per_launch_raw": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
       "fetch_bytes_per_launch_corrected_x2": 2 * fetch_kb * 1024, "write_bytes_per_launch": write_kb * 1024,
       "traffic_bytes_per_launch": round(2 * fetch_kb * 1024 + write_kb * 1024),
       "method": "rocprofv3 --kernel-trace --pmc <one counter per pass> -- python3 bench.py --replay-dominant-only; "
                 "averages over the launches of one iteration's mix (6 x T=6016 d=100, 4 x T=3008 d=100, 2 x T=3008 d=512), "
                 "warm-up pass included; FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B)"}
json.dump(out, open("gpurun_out/r03_wgrad_traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
