# rocprofv3 counter passes over a lab script: bash tools/lab/pmc_any.sh <script.py> <kernel-name-substring>
R=$PWD
SC=$1; PAT=$2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc1 -- python3 $R/$SC > $R/gpurun_out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/pmc2 -- python3 $R/$SC > $R/gpurun_out/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/pmc3 -- python3 $R/$SC > $R/gpurun_out/pmc3.log 2>&1
cd $R
python3 - "$PAT" <<'PY'
import csv, glob, collections, sys
pat = sys.argv[1]
for d in ("pmc1", "pmc2", "pmc3"):
    fs = glob.glob("gpurun_out/%s/*/*counter_collection.csv" % d)
    if not fs:
        print(d, "no counter file", glob.glob("gpurun_out/%s/*/*" % d)); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if pat not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("ganffn::", "").split("(")[0][:50] + " g" + r["Grid_Size"]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in sorted(agg.items()):
        print(d, k, {n: round(sum(v) / len(v)) for n, v in c.items()})
PY
rm -rf gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc3
