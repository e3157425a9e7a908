#!/usr/bin/env python3
"""rocprofv3 PMC passes of tools/pmc_lba.sh -> one JSON: per kernel of the window-batched local BA (the LARGEST launches of a run = the full trial
slots, all windows active) the counters, the duration of the same launches in the --stats pass, and what follows from them:
    mfma_busy_frac   = SQ_VALU_MFMA_BUSY_CYCLES / (duration * clock * SIMDs)          (the matrix pipes' share of the kernel)
    mfma_flops       = SQ_INSTS_VALU_MFMA_MOPS_F64 * 512                              (rocprofv3's own MfmaFlopsF64)
    mfma_util        = mfma_flops / duration / 78.6e12                                 (against the f64 matrix peak, MI355X_MICROARCH.md)
usage: pmc_lba_summary.py <gpurun_out> <tag 16|1> <out.json>"""
import collections, csv, glob, json, os, re, sys

SIMDS, PEAK = 256 * 4, 78.6e12


def newest(pat):
    return max(glob.glob(pat), key=os.path.getmtime)


def counters(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(newest(d + "/*/*counter_collection.csv"))):
        name = re.sub(r"<.*?>", "", r["Kernel_Name"].split("(")[0]).replace("void ", "").replace("rumi::", "").strip()
        agg[name][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return agg


def durations(d):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(newest(d + "/*/*kernel_trace.csv"))):
        name = re.sub(r"<.*?>", "", r["Kernel_Name"].split("(")[0]).replace("void ", "").replace("rumi::", "").strip()
        agg[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
    return agg


def top(vals, frac=0.5):       # the full-size launches: the upper half by value (init / exiting slots are much smaller)
    v = sorted(vals, reverse=True)
    v = v[:max(1, int(len(v) * frac))]
    return sum(v) / len(v)


if __name__ == "__main__":
    root, tag, out = sys.argv[1], sys.argv[2], sys.argv[3]
    res = {"windows": int(tag), "note": "averages over the larger half of a kernel's launches (full LM trial slots); counters and durations from separate passes of the same command"}
    dur = durations(root + "/pmc_lba_time" + tag)
    ks = {}
    for sub in ("mfma", "sq"):
        for name, cs in counters(root + "/pmc_lba_%s%s" % (sub, tag)).items():
            if not name.startswith("k_baw"): continue
            k = ks.setdefault(name, {})
            for cn, vals in cs.items():
                k[cn] = top([v for _, v in vals])
    for name, k in ks.items():
        if name in dur: k["duration_us"] = round(top(dur[name]) * 1e6, 2)
        if "GRBM_GUI_ACTIVE" in k and "duration_us" in k:
            k["clock_ghz_from_grbm"] = round(k["GRBM_GUI_ACTIVE"] / 8 / (k["duration_us"] * 1e-6) / 1e9, 3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in k and "SQ_BUSY_CYCLES" in k and k["SQ_BUSY_CYCLES"] > 0:
            k["mfma_busy_over_sq_busy"] = round(k["SQ_VALU_MFMA_BUSY_CYCLES"] / k["SQ_BUSY_CYCLES"], 4)
        if "SQ_INSTS_VALU_MFMA_MOPS_F64" in k and "duration_us" in k:
            k["mfma_flops"] = k["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512
            k["mfma_util_vs_78.6TF"] = round(k["mfma_flops"] / (k["duration_us"] * 1e-6) / PEAK, 4)
        if "SQ_ACTIVE_INST_VALU" in k and "duration_us" in k:
            k["valu_busy_frac_at_2.4GHz"] = round(k["SQ_ACTIVE_INST_VALU"] * 4 / (k["duration_us"] * 1e-6 * 2.4e9 * SIMDS), 4)
    res["kernels"] = ks
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))
