#!/usr/bin/env python3
"""Concurrency picture of a rocprofv3 --kernel-trace run: per kernel name the summed duration, and over the busiest window the time with
0 / 1 / 2 / >= 3 kernels in flight.  usage: trace_overlap.py <dir>"""
import collections, csv, glob, os, re, sys
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(f))]
ev, per = [], collections.defaultdict(lambda: [0, 0])
for r in rows:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()[:40]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    per[n][0] += e - s; per[n][1] += 1
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
# the middle of the run = steady state (run the bench with many steps)
lo, hi = t0 + (t1 - t0) * 0.45, t0 + (t1 - t0) * 0.8
depth, last, hist = 0, None, collections.Counter()
for t, d in ev:
    if last is not None and t > lo and last < hi:
        hist[min(depth, 4)] += min(t, hi) - max(last, lo)
    depth += d; last = t
tot = sum(hist.values())
print("window %.2f ms; kernels in flight: " % (tot / 1e6) + ", ".join("%d: %.1f %%" % (k, 100 * v / tot) for k, v in sorted(hist.items())))
for n, (d, c) in sorted(per.items(), key=lambda x: -x[1][0])[:14]:
    print("  %-42s %6d calls  %9.3f ms total  %8.1f us avg" % (n, c, d / 1e6, d / c / 1e3))
