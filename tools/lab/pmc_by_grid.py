import csv, glob, sys, collections
for counter in sys.argv[1:]:
    f = glob.glob("gpurun_out/pmc_%s/*/*counter_collection.csv" % counter)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and ("gemm" in r["Kernel_Name"]):
            agg[(r["Kernel_Name"][:60], r["Grid_Size"])].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(counter, k, "n=%d avg=%.1f MB" % (len(v), sum(v) / len(v) / 1024))
