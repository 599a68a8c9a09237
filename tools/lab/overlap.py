"""lab: how much of the multi-stream step do kernels of different streams actually share the chip?
    python tools/lab/overlap.py <kernel_trace.csv> [skip_fraction]
From the begin / end stamps of a rocprofv3 kernel trace: the wall span, the time with 0 / 1 / 2 / 3+ kernels in flight, and per
kernel family its summed duration, its `solo` time (nothing else in flight) and the mean number of kernels in flight while it
runs.  The leading skip_fraction of the trace (warm-up, stream tuning) is dropped."""
import collections
import csv
import sys


def short(n):
    return n.replace('(anonymous namespace)::', '').replace('ganffn::', '').replace('void ', '').split('(')[0]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r.get('Queue_Id', '?')) for r in rows))
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    cut = t0 + skip * (t1 - t0)
    ev = [e for e in ev if e[0] >= cut]
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    pts = []
    for i, (b, e, n, q) in enumerate(ev):
        pts.append((b, 1, i))
        pts.append((e, -1, i))
    pts.sort()
    level_time = collections.Counter()
    solo = collections.Counter()
    dur = collections.Counter()
    conc = collections.Counter()          # integral of (kernels in flight) over the kernel's own span
    live = set()
    prev = pts[0][0]
    for t, d, i in pts:
        dt = t - prev
        if dt > 0:
            level_time[min(len(live), 4)] += dt
            for j in live:
                conc[ev[j][2]] += dt * len(live)
            if len(live) == 1:
                solo[ev[next(iter(live))][2]] += dt
        prev = t
        if d == 1:
            live.add(i)
        else:
            live.discard(i)
    for b, e, n, q in ev:
        dur[n] += e - b
    span = t1 - t0
    print("span %.1f ms, %d kernels, queues %s" % (span / 1e6, len(ev), sorted(set(e[3] for e in ev))))
    for k in sorted(level_time):
        print("  %d%s kernels in flight: %6.2f ms  %5.1f%%" % (k, "+" if k == 4 else " ", level_time[k] / 1e6, 100.0 * level_time[k] / span))
    print("sum of kernel durations %.1f ms = %.2f x span" % (sum(dur.values()) / 1e6, sum(dur.values()) / span))
    print("%-44s %9s %9s %6s" % ("kernel", "sum ms", "solo ms", "conc"))
    for n, d in sorted(dur.items(), key=lambda kv: -kv[1])[:24]:
        print("%-44s %9.2f %9.2f %6.2f" % (n[:44], d / 1e6, solo[n] / 1e6, conc[n] / d))


if __name__ == "__main__":
    main()
