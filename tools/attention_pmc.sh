#!/bin/bash
# north_star's "MFMA utilisation on the attention GEMMs against gfx950 peak": SQ counters of the attention kernels over one
# iteration's launch mix (bench.py --replay-family attention: attn16_fwd/bwd<10,6,..> at 320 and 640 (dialogue, head)
# problems, attention_fwd/bwd<64,3> at 256), rocprofv3 --pmc with --kernel-trace only (one counter set per pass, program
# directly after `--`).  Run from the repo root on the GPU box; writes gpurun_out/r05_attention_pmc.json.
set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_attn1 $R/gpurun_out/pmc_attn2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_attn1 -- python3 $R/bench.py --replay-family attention > $R/gpurun_out/pmc_attn1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/pmc_attn2 -- python3 $R/bench.py --replay-family attention > $R/gpurun_out/pmc_attn2.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, json, collections, sys
sys.path.insert(0, "tools")
import roofline_model as RM
CLK, SIMDS = 2.4e9, 1024                      # nominal shader clock, 256 CUs x 4 SIMDs
S, E10, E64 = 94, 100, 512
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in ("pmc_attn1", "pmc_attn2"):
    f = glob.glob("gpurun_out/%s/*/*counter_collection.csv" % d)
    if not f:
        print("no counter file in", d, glob.glob("gpurun_out/%s/*/*" % d)); continue
    seen = set()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("ganffn::", "").replace("void ", "").split("(")[0]
        if "att" not in k:
            continue
        key = "%s grid %d" % (k, int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if d == "pmc_attn1" and (r["Dispatch_Id"]) not in seen and "Start_Timestamp" in r:
            seen.add(r["Dispatch_Id"])
            dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {"what": "SQ counters per launch (averages over the launches of one iteration's attention mix, bench.py --replay-family attention), "
               "S = 94; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (launch duration x 2.4 GHz x 1024 SIMDs) = the share of the chip's MFMA "
               "issue slots the launch used (padded MFMAs included: head_dim 10 fills 10 of 16 tile rows / 10 of 12 k); "
               "useful_frac_of_fp32_mfma_peak = algorithmic FLOPs (4 S hd per (token, head) forward, 2.5 x backward) / duration / 157.3 TFLOP/s",
       "csrc_sha16": RM.csrc_sha16(), "kernels": {}}
for key, c in sorted(agg.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    us = sum(dur[key]) / len(dur[key]) if dur.get(key) else None
    probs = int(key.split("grid ")[1])
    k = key.split(" grid")[0]
    hd = 64 if "<64" in k else 10
    if "attn16_fwd" in k and "<10, 6, 2>" in k:
        probs //= 3                                   # the forward cuts a (dialogue, head) problem into 3 workgroups
    flop = 4.0 * S * hd * S * probs * (2.5 if "bwd" in k else 1.0)
    row = {"launches": len(next(iter(c.values()))), "avg_us_under_pmc": round(us, 2) if us else None, "problems": probs}
    row.update({n: round(v) for n, v in m.items()})
    if us:
        row["mfma_busy_frac"] = round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (us * 1e-6 * CLK * SIMDS), 4)
        row["useful_frac_of_fp32_mfma_peak"] = round(flop / (us * 1e-6) / RM.FP32_MFMA_PEAK, 4)
    if m.get("SQ_WAVE_CYCLES"):
        for n in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"):
            if n in m:
                row[n + "_frac_of_wave_cycles"] = round(m[n] / m["SQ_WAVE_CYCLES"], 4)
    out["kernels"][key] = row
json.dump(out, open("gpurun_out/r05_attention_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
PY
rm -rf gpurun_out/pmc_attn1 gpurun_out/pmc_attn2
