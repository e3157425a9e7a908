"""Timeline of the last `span_ms` of a rocprofv3 run with --kernel-trace --memory-copy-trace: kernels and copies merged by start time."""
import csv, glob, os, sys
d, span_ms = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = []
for f in glob.glob(d + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:]))
for f in glob.glob(d + "/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
end = max(e[1] for e in ev)
t0 = end - int(span_ms * 1e6)
prev = None
for s, e, n in ev:
    if e < t0: continue
    gap = (s - prev) / 1e3 if prev else 0
    print("%9.1f us  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, n))
    prev = e
