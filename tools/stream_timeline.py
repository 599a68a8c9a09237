"""Per-queue (HIP stream) busy time and concurrency of a rocprofv3 kernel trace CSV restricted to the timed region
(the last N iterations are not separable in the trace, so the whole run is summarised)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
by_q = collections.defaultdict(list)
for r in rows:
    q = r.get("Queue_Id") or r.get("Stream_Id") or "?"
    by_q[q].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
t0 = min(s for v in by_q.values() for s, _, _ in v)
t1 = max(e for v in by_q.values() for _, e, _ in v)
print("span %.1f ms, %d kernels, %d queues" % ((t1 - t0) / 1e6, len(rows), len(by_q)))
for q, v in sorted(by_q.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
    busy = sum(e - s for s, e, _ in v)
    print("queue %s: %6d kernels, busy %8.1f ms (%.0f%% of span), first %.1f ms, last %.1f ms" %
          (q, len(v), busy / 1e6, 100.0 * busy / (t1 - t0), (min(s for s, _, _ in v) - t0) / 1e6, (max(e for _, e, _ in v) - t0) / 1e6))
# concurrency histogram
ev = []
for v in by_q.values():
    for s, e, _ in v:
        ev.append((s, 1)); ev.append((e, -1))
ev.sort()
cur, last, hist = 0, ev[0][0], collections.Counter()
for t, d in ev:
    hist[cur] += t - last
    cur += d
    last = t
tot = sum(hist.values())
print("kernels in flight: " + ", ".join("%d: %.0f%%" % (k, 100.0 * hist[k] / tot) for k in sorted(hist)))
