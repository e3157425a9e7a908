#!/usr/bin/env python3
"""Per-kernel averages of every counter in the newest rocprofv3 counter_collection.csv under the given directories.
usage: pmc_kernel.py <dir> [<dir> ...] [--like substr]"""
import collections, csv, glob, os, re, sys
dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
like = sys.argv[sys.argv.index("--like") + 1] if "--like" in sys.argv else "rumi::"
if "--like" in sys.argv: dirs.remove(like)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    f = max(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
        if like in n:
            agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n in sorted(agg):
    print(n, " ".join("%s=%.4g(n%d)" % (c, sum(v) / len(v), len(v)) for c, v in sorted(agg[n].items())))
