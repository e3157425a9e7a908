"""Timeline of the LAST n ms of a rocprofv3 *_kernel_trace.csv: start offset (us), duration, queue, kernel."""
import csv, sys
f, span_ms = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
end = max(int(r["End_Timestamp"]) for r in rows)
t0 = end - int(span_ms * 1e6)
qs = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e < t0: continue
    q = qs.setdefault(r["Queue_Id"], len(qs))
    print("%9.1f %7.1f  q%d  %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, q, "    " * q, r["Kernel_Name"].split("(")[0][-40:]))
