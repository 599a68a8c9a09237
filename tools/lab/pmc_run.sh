# rocprofv3 counter passes over tools/lab/kern_pmc.py (one --pmc set per pass; no trace domains besides kernel-trace)
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc1 -- python3 $R/tools/lab/kern_pmc.py > $R/gpurun_out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/pmc2 -- python3 $R/tools/lab/kern_pmc.py > $R/gpurun_out/pmc2.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for d in ("pmc1", "pmc2"):
    fs = glob.glob("gpurun_out/%s/*/*counter_collection.csv" % d)
    if not fs:
        print(d, "no counter file", glob.glob("gpurun_out/%s/*/*" % d)); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].replace("ganffn::", "").split("(")[0][:60] + " g" + r["Grid_Size"]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in agg.items():
        if "gemm" in k or "attn" in k:
            print(d, k, {n: round(sum(v) / len(v)) for n, v in c.items()})
PY
rm -rf gpurun_out/pmc1 gpurun_out/pmc2
