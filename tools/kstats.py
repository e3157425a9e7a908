"""Print a rocprofv3 *_kernel_stats.csv compactly (name truncated, calls, total us, average us, min, max)."""
import csv, sys
for f in sys.argv[1:]:
    print(f)
    for r in csv.DictReader(open(f)):
        n = r["Name"].split("(")[0][-60:]
        print("  %-60s calls %6s total_us %10.1f avg_us %8.2f min %8.2f max %8.2f" % (n, r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
