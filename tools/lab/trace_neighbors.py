"""print the kernels launched right before / after each occurrence of a given kernel-name substring in a rocprofv3 kernel trace
(single-stream run): usage trace_neighbors.py <kernel_trace.csv> <substring>"""
import collections, csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('ganffn::', '').replace('void ', '').split('(')[0]
    rows.append((int(r['Start_Timestamp']), n, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
rows.sort()
rows = rows[len(rows) // 2:]
ctx = collections.Counter()
for i, (t, n, d) in enumerate(rows):
    if sys.argv[2] in n and 0 < i < len(rows) - 1:
        ctx[(rows[i - 1][1][:60], rows[i + 1][1][:60])] += 1
for (a, b), c in ctx.most_common(15):
    print("%5d  after %-60s before %s" % (c, a, b))
