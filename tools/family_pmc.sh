#!/bin/bash
# SQ (+ TCC write-side) counters of one kernel family's launch mix, as a JSON artefact:
#     bash tools/family_pmc.sh ffn_k100|ffn_n100|attention|wgrad|gemm_generic [ROUND]
# -> gpurun_out/<ROUND>_<family>_pmc.json (ROUND defaults to r05).  bench.py --replay-family <family> replays one
# iteration's launches of that family exactly as the encoder stack issues them; rocprofv3 --pmc with --kernel-trace only,
# one counter set per pass, the program directly after `--` (MI355X_MICROARCH.md "rocprofv3 PMC slots": 8 SQ slots per pass;
# FETCH_SIZE and WRITE_SIZE in passes of their own).  Run from the repo root on the GPU box.
set -e
FAM=${1:-ffn_k100}
ROUND=${2:-r05}
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmcf1 $R/gpurun_out/pmcf2 $R/gpurun_out/pmcf3 $R/gpurun_out/pmcf4
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmcf1 -- python3 $R/bench.py --replay-family $FAM > $R/gpurun_out/pmcf1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/pmcf2 -- python3 $R/bench.py --replay-family $FAM > $R/gpurun_out/pmcf2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmcf3 -- python3 $R/bench.py --replay-family $FAM > $R/gpurun_out/pmcf3.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmcf4 -- python3 $R/bench.py --replay-family $FAM > $R/gpurun_out/pmcf4.log 2>&1
cd $R
python3 - $FAM $ROUND <<'PY'
import csv, glob, json, collections, sys
sys.path.insert(0, "tools")
import roofline_model as RM
fam, rnd = sys.argv[1], sys.argv[2]
CLK, SIMDS = 2.4e9, 1024                      # nominal shader clock, 256 CUs x 4 SIMDs
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in ("pmcf1", "pmcf2", "pmcf3", "pmcf4"):
    f = glob.glob("gpurun_out/%s/*/*counter_collection.csv" % d)
    if not f:
        print("no counter file in", d, glob.glob("gpurun_out/%s/*/*" % d)); continue
    seen = set()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("ganffn::", "").replace("void ", "").split("(")[0]
        if not any(x in k for x in ("gemm", "tn100", "attn", "attention", "rc_")):
            continue
        key = "%s grid %d" % (k[:80], int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if d == "pmcf1" and r["Dispatch_Id"] not in seen and "Start_Timestamp" in r:
            seen.add(r["Dispatch_Id"])
            dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {"what": "counters per launch (averages over the launches of one iteration's launch mix of the family, bench.py --replay-family %s), "
               "S = 94, B = 32.  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (launch duration x 2.4 GHz x 1024 SIMDs) = the share of the "
               "chip's MFMA issue slots the launch used; *_frac_of_wave_cycles: SQ_WAIT_ANY = parked in s_waitcnt / barrier, "
               "SQ_WAIT_INST_ANY = issue stall (MFMA RAW / pipe busy), SQ_ACTIVE_INST_ANY = issuing (disjoint, sum ~ 1); "
               "valu_per_mfma = SQ_INSTS_VALU (MFMAs included) / SQ_INSTS_MFMA; hbm_write_bytes = WRITE_SIZE x 1024 (KB -> B), "
               "hbm_read_bytes = FETCH_SIZE x 1024 x 2 (gfx950 correction, MI355X_MICROARCH.md)" % fam,
       "family": fam, "csrc_sha16": RM.csrc_sha16(), "kernels": {}}
for key, c in sorted(agg.items(), key=lambda kv: -sum(dur.get(kv[0], [0]))):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    us = sum(dur[key]) / len(dur[key]) if dur.get(key) else None
    if us is None or us < 3:
        continue
    row = {"launches": len(dur[key]), "avg_us_under_pmc": round(us, 2)}
    row.update({n: round(v) for n, v in m.items()})
    row["mfma_busy_frac"] = round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (us * 1e-6 * CLK * SIMDS), 4)
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        for n in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS"):
            if n in m:
                row[n + "_frac_of_wave_cycles"] = round(m[n] / wc, 4)
        if m.get("SQ_BUSY_CYCLES"):
            row["avg_waves_per_simd_while_busy"] = round(wc / m["SQ_BUSY_CYCLES"] / 4.0, 3)    # SQ_BUSY_CYCLES counts per SE-level SQ; indicative only
    if m.get("SQ_INSTS_MFMA"):
        row["valu_per_mfma"] = round(m.get("SQ_INSTS_VALU", 0) / m["SQ_INSTS_MFMA"], 3)
        row["lds_conflict_cycles_per_mfma"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_INSTS_MFMA"], 3)
    if "WRITE_SIZE" in m:
        row["hbm_write_bytes"] = round(m["WRITE_SIZE"] * 1024)
    if "FETCH_SIZE" in m:
        row["hbm_read_bytes"] = round(m["FETCH_SIZE"] * 1024 * 2)
    out["kernels"][key] = row
path = "gpurun_out/%s_%s_pmc.json" % (rnd, fam)
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1)[:8000])
PY
rm -rf gpurun_out/pmcf1 gpurun_out/pmcf2 gpurun_out/pmcf3 gpurun_out/pmcf4
