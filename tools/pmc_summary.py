#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md §HBM prescribes) of
`bench.py --steps 2 --warmup 1 --no-cpu` into profiles/<tag>_pmc_traffic.json: HBM bytes per launch and kernel,
    bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
The factor 2 on the read side is the gfx950 correction of the guide, re-calibrated here with tools/pmc_calib.hip for BOTH
16 B/lane and 4 B/lane coalesced streaming reads of 1 GiB (FETCH_SIZE * 1024 = 0.5000 x bytes read in both cases).
usage: pmc_summary.py <dir-with-FETCH_SIZE-run> <dir-with-WRITE_SIZE-run> <out.json> [frames_per_launch]"""
import collections, csv, glob, json, os, re, sys


def per_kernel(d, counter):
    f = max(glob.glob(d + "/*/*counter_collection.csv"), key=os.path.getmtime)     # gpurun merges every run into the same directory
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = re.sub(r"<.*?>", "", r["Kernel_Name"].split("(")[0]).replace("void ", "").strip()   # template arguments / return type dropped
        if name.startswith("rumi::") and r["Counter_Name"] == counter:
            agg[name.replace("rumi::", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


if __name__ == "__main__":
    fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
    write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
    frames = int(sys.argv[4]) if len(sys.argv) > 4 else 256
    out = {"unit": "bytes per launch", "frames_per_launch": frames,
           "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024; read-side factor 2 calibrated with tools/pmc_calib.hip (4 B/lane and 16 B/lane)",
           "kernels": {k: {"fetch_kb": round(fetch[k], 1), "write_kb": round(write.get(k, 0.0), 1),
                           "hbm_bytes_per_launch": round((2 * fetch[k] + write.get(k, 0.0)) * 1024), "launches_sampled": nf[k]}
                       for k in sorted(fetch)}}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1))
