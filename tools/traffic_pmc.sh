#!/bin/bash
# HBM-side traffic of a kernel family's launch mix, per MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE in SEPARATE
# rocprofv3 --pmc passes (kernel trace only), FETCH_SIZE doubled on gfx950.  Run from the repo root on the GPU box:
#   bash tools/traffic_pmc.sh wgrad          -> gpurun_out/r05_wgrad_traffic.json
#   bash tools/traffic_pmc.sh gemm_generic   -> gpurun_out/r05_gemm_traffic.json
# (copied into profiles/ afterwards; the raw per-dispatch CSVs stay under gpurun_out/).
set -e
FAM=${1:-wgrad}
R=$PWD
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_${FAM}_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_${FAM}_$c -- python3 $R/bench.py --replay-family $FAM > $R/gpurun_out/pmc_${FAM}_$c.log 2>&1
done
cd $R
python3 - $FAM <<'PY'
import csv, glob, json, sys
fam = sys.argv[1]
sys.path.insert(0, "tools")
import roofline_model as RM
table = {f["key"]: f for f in RM.family_table(94, 32)}[fam]
def per_launch(counter):
    # bytes summed over the family's kernels, divided by its LOGICAL launches (a reduce kernel belongs to the launch before it)
    f = glob.glob("gpurun_out/pmc_%s_%s/*/*counter_collection.csv" % (fam, counter))[0]
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("ganffn::", "").replace("void ", "")
        if any(k.startswith(p) for p in table["prefixes"]):
            tot += float(r["Counter_Value"])
            if "reduce" not in k:
                n += 1
    return tot / n, n
fetch_kb, n1 = per_launch("FETCH_SIZE")
write_kb, n2 = per_launch("WRITE_SIZE")
out = {"family": fam, "kernel": table["title"], "seq_len": 94, "dialogues_per_gpu": 32, "dispatches_profiled": [n1, n2],
       "FETCH_SIZE_KB_per_launch_raw": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
       "fetch_bytes_per_launch_corrected_x2": 2 * fetch_kb * 1024, "write_bytes_per_launch": write_kb * 1024,
       "traffic_bytes_per_launch": round(2 * fetch_kb * 1024 + write_kb * 1024), "csrc_sha16": RM.csrc_sha16(),
       "method": "rocprofv3 --kernel-trace --pmc <one counter per pass> -- python3 bench.py --replay-family %s; averages over the "
                 "launches of one iteration's mix, warm-up pass included; FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B)" % fam}
name = {"wgrad": "r05_wgrad_traffic.json", "gemm_generic": "r05_gemm_traffic.json"}.get(fam, "r05_%s_traffic.json" % fam)
json.dump(out, open("gpurun_out/" + name, "w"), indent=1)
print(json.dumps(out))
PY
