"""Summarise a rocprofv3 kernel trace CSV by (kernel, grid): share, launches, avg us."""
import collections
import csv
import sys

rows = csv.DictReader(open(sys.argv[1]))
agg = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('ganffn::', '').replace('void ', '').split('(')[0]
    key = (n, int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), r['Grid_Size_Y'], r['Grid_Size_Z'])
    agg[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in agg.values())
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
if len(sys.argv) > 3:
    print("iterations %d" % int(sys.argv[3]))         # step iterations the trace covers (warm-up + timed), for per-iteration figures
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from roofline_model import csrc_sha16
    print("csrc_sha16 %s" % csrc_sha16())             # the kernel sources this trace was taken on (bench.py flags a stale profile)
print("share  launches  avg_us  grid(blocks_x,y,z)  kernel")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print("%5.2f%% %6d %8.1f  (%s,%s,%s)  %s" % (100 * sum(v) / tot, len(v), sum(v) / len(v), k[1], k[2], k[3], k[0]))
print("total GPU kernel time %.1f ms" % (tot / 1e3))
byname = collections.defaultdict(float)
for k, v in agg.items():
    byname[k[0]] += sum(v)
print("--- by kernel")
for n, t in sorted(byname.items(), key=lambda kv: -kv[1])[:20]:
    print("%5.2f%%  %s" % (100 * t / tot, n))
