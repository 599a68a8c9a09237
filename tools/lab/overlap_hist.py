"""How many kernels are in flight over time in a rocprofv3 kernel trace (CSV) of the multi-stream step: the share of wall time
with 0 / 1 / 2 / 3+ kernels running, per-queue busy share, and the kernels that run ALONE the longest (the serial parts of
the sub-step DAG).  Looks at the last 60 % of the trace (past the warm-up).  usage: overlap_hist.py <kernel_trace.csv>"""
import collections, csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('ganffn::', '').replace('void ', '').split('(')[0]
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '?'), n))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t0 + 0.4 * (t1 - t0)
rows = [r for r in rows if r[0] >= lo]
ev = []
for i, (s, e, q, n) in enumerate(rows):
    ev.append((s, 1, i)); ev.append((e, -1, i))
ev.sort()
hist = collections.Counter(); alone = collections.Counter(); active = set(); last = ev[0][0]
for t, d, i in ev:
    dt = t - last
    if dt > 0:
        hist[min(len(active), 4)] += dt
        if len(active) == 1:
            alone[rows[next(iter(active))][3]] += dt
    last = t
    if d == 1: active.add(i)
    else: active.discard(i)
tot = sum(hist.values())
print("wall %.1f ms; kernels in flight: " % (tot / 1e6) + ", ".join("%d%s: %.1f %%" % (k, "+" if k == 4 else "", 100 * v / tot) for k, v in sorted(hist.items())))
busy = collections.Counter()
for s, e, q, n in rows:
    busy[q] += e - s
print("busy share per queue: " + ", ".join("q%s %.1f %%" % (q, 100 * b / tot) for q, b in sorted(busy.items())))
print("time with exactly ONE kernel in flight, by kernel (top 12):")
for n, t in alone.most_common(12):
    print("  %5.2f %% of wall  %s" % (100 * t / tot, n))
