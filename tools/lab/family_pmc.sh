#!/bin/bash
# SQ counters (two passes) of one kernel family's launch mix: bash tools/lab/family_pmc.sh wgrad|gemm_generic|attention
# -> gpurun_out/r4_pmc_<family>.txt (per (kernel, grid): averages per launch, MFMA-busy share of the chip's issue slots)
FAM=${1:-wgrad}
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmcf1 $R/gpurun_out/pmcf2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmcf1 -- python3 $R/bench.py --replay-family $FAM > $R/gpurun_out/pmcf1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/pmcf2 -- python3 $R/bench.py --replay-family $FAM > $R/gpurun_out/pmcf2.log 2>&1
cd $R
python3 - $FAM <<'PY' | tee gpurun_out/r4_pmc_$FAM.txt
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for d in ("pmcf1", "pmcf2"):
    f = glob.glob("gpurun_out/%s/*/*counter_collection.csv" % d)
    if not f: print("no counters in", d); continue
    seen = set()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("ganffn::", "").replace("void ", "").split("(")[0]
        if not any(x in k for x in ("gemm", "tn", "attn", "attention")): continue
        key = "%s grid %d" % (k[:60], int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if d == "pmcf1" and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for key, c in sorted(agg.items(), key=lambda kv: -sum(dur.get(kv[0], [0]))):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    us = sum(dur[key]) / len(dur[key]) if dur.get(key) else 0
    if us < 5: continue
    wc = m.get("SQ_WAVE_CYCLES", 1)
    print("%-70s %4d x %8.1f us | mfma busy %.3f of issue slots | wave cycles: active %.2f wait_inst %.2f wait_any %.2f | lds insts %d conflict cycles %d (%.2f per inst) wait_inst_lds %.3f" % (
        key, len(dur[key]), us, m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (us * 1e-6 * 2.4e9 * 1024), m.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        m.get("SQ_WAIT_INST_ANY", 0) / wc, m.get("SQ_WAIT_ANY", 0) / wc, m.get("SQ_INSTS_LDS", 0), m.get("SQ_LDS_BANK_CONFLICT", 0),
        m.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, m.get("SQ_INSTS_LDS", 1)), m.get("SQ_WAIT_INST_LDS", 0) / wc))
PY
rm -rf gpurun_out/pmcf1 gpurun_out/pmcf2
