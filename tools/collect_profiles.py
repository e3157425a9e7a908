#!/usr/bin/env python3
"""Copies the newest rocprofv3 outputs of a measurement pass from gpurun_out/ into profiles/ (round tag: first argument, default r04) and prints a summary.
Expects gpurun_out/{prof_default,prof_serial,prof_lba,pmc_fetch,pmc_write,pmc_a,pmc_b}, bench_r02.json, stage_serial.log, host_batch.log, r02_valu_issue_rates.txt."""
import collections, csv, glob, json, os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r04"


def newest(pattern):
    return max(glob.glob(os.path.join(G, pattern)), key=os.path.getmtime)


shutil.copy(newest("prof_default/*/*kernel_stats.csv"), os.path.join(P, TAG + "_extract_match_kernel_stats.csv"))
shutil.copy(newest("prof_serial/*/*kernel_stats.csv"), os.path.join(P, TAG + "_extract_match_serial_kernel_stats.csv"))
try:
    shutil.copy(newest("prof_serial128/*/*kernel_stats.csv"), os.path.join(P, TAG + "_extract_match_serial_128frames_kernel_stats.csv"))
    shutil.copy(os.path.join(G, "bench_%s_records.json" % TAG), os.path.join(P, TAG + "_bench_line_records_path.json"))
except (ValueError, OSError):
    print("missing the 128-frame serial pass / the records-path bench line")
shutil.copy(newest("prof_lba/*/*kernel_stats.csv"), os.path.join(P, TAG + "_lba_20kf_3000mp_kernel_stats.csv"))
# second pass (tools/measure_more.sh): latency paths
for src, dst in (("prof_track/*/*kernel_stats.csv", TAG + "_track_frame_kernel_stats.csv"), ("prof_single/*/*kernel_stats.csv", TAG + "_single_frame_kernel_stats.csv"),
                 ("pmc_lba_time16/*/*kernel_stats.csv", TAG + "_lba_batch16_kernel_stats.csv"), ("pmc_lba_time1/*/*kernel_stats.csv", TAG + "_lba_single_window_kernel_stats.csv")):
    try:
        shutil.copy(newest(src), os.path.join(P, dst))
    except ValueError:
        print("missing", src)
for src, dst in (("track_probe.log", TAG + "_track_frame_probe.txt"), ("lba_probe.log", TAG + "_lba_probe.txt"), ("bow_batch.log", TAG + "_bow_batch_probe.txt"),
                 ("pose_probe.log", TAG + "_pose_probe.txt"), ("rsq_probe.log", TAG + "_rsq_rcp_accuracy.txt"), ("bench_matrix.json", TAG + "_bench_matrix.json"),
                 ("single_frame.log", TAG + "_single_frame_probe.txt"), ("queue_probe.log", TAG + "_queue_probe.txt"), ("bench_one_process.json", TAG + "_bench_line_one_process_queue.json"),
                 ("test_queue_cc.log", TAG + "_queue_c_test.txt")):
    if os.path.exists(os.path.join(G, src)):
        txt = open(os.path.join(G, src)).read()
        open(os.path.join(P, dst), "w").write("\n".join(l for l in txt.splitlines() if "amdgpu.ids" not in l) + "\n")
    else:
        print("missing", src)
try:
    res = {}
    for tag, key in (("16", "batch16"), ("1", "single")):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_lba_summary.py"), G, tag, "/tmp/_lba_%s.json" % tag], stdout=subprocess.DEVNULL)
        res[key] = json.load(open("/tmp/_lba_%s.json" % tag))
    res["command"] = "tools/pmc_lba.sh: RUMI_BAW_GROUPS=1 rocprofv3 --kernel-trace --pmc <counters> -- python3 tools/prof_lba_batch.py {16|0} 2 (separate passes for the MFMA counters, the SQ counters and --stats)"
    json.dump(res, open(os.path.join(P, TAG + "_lba_pmc_mfma.json"), "w"), indent=1)
except Exception as e:
    print("local BA counter passes missing:", e)
for d, tmp in (("pmc_fetch", "/tmp/_pf"), ("pmc_write", "/tmp/_pw")):
    shutil.rmtree(tmp, ignore_errors=True); os.makedirs(tmp + "/x")
    shutil.copy(newest(d + "/*/*counter_collection.csv"), tmp + "/x/")
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), "/tmp/_pf", "/tmp/_pw", os.path.join(P, TAG + "_pmc_traffic.json"), "256"],
                      stdout=subprocess.DEVNULL)
out = {"command": "RUMI_SERIAL=1 rocprofv3 --kernel-trace --pmc <8 SQ counters> -- python3 bench.py --steps 2 --warmup 1 --batch 256 --no-cpu (two passes)",
       "unit": "counter value per launch (256 frames), averaged over the sampled launches", "kernels": {}}
for d in ("pmc_a", "pmc_b"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(newest(d + "/*/*counter_collection.csv"))):
        n = re.sub(r"<.*?>", "", r["Kernel_Name"].split("(")[0]).replace("void ", "").strip()   # template arguments / return type dropped
        if n.startswith("rumi::"):
            agg[n.replace("rumi::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n, c in agg.items():
        out["kernels"].setdefault(n, {}).update({k: round(sum(v) / len(v)) for k, v in c.items()})
json.dump(out, open(os.path.join(P, TAG + "_pmc_sq_counters.json"), "w"), indent=1)
open(os.path.join(P, TAG + "_bench_line.json"), "w").write([l for l in open(os.path.join(G, "bench_%s.json" % TAG)).read().splitlines() if l.startswith("{")][-1] + "\n")
shutil.copy(os.path.join(G, "host_batch.log"), os.path.join(P, TAG + "_host_batch_probe.txt"))
shutil.copy(os.path.join(G, TAG + "_valu_issue_rates.txt"), os.path.join(P, TAG + "_valu_issue_rates.txt"))
shutil.copy(os.path.join(G, "stage_serial.log"), os.path.join(P, TAG + "_stage_ms_standalone.txt"))
for f in (TAG + "_extract_match_kernel_stats.csv", TAG + "_extract_match_serial_kernel_stats.csv", TAG + "_lba_20kf_3000mp_kernel_stats.csv"):
    print(f)
    for r in list(csv.DictReader(open(os.path.join(P, f))))[:12]:
        print("   ", r["Name"][:42].ljust(44), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us", r["Percentage"])
t = json.load(open(os.path.join(P, TAG + "_pmc_traffic.json")))["kernels"]
for k, v in out["kernels"].items():
    print(k.ljust(16), "VALU", round(v.get("SQ_INSTS_VALU", 0) / 1e6, 1), "M  LDS", round(v.get("SQ_INSTS_LDS", 0) / 1e6, 1), "M  HBM", round(t.get(k, {}).get("hbm_bytes_per_launch", 0) / 1e6, 1), "MB")
b = json.load(open(os.path.join(P, TAG + "_bench_line.json")))
print({k: b.get(k) for k in ("value", "value_h2d_inclusive", "single_frame_host_api_fps", "batch_sweep_fps", "ms_per_step", "roofline", "stage_ms_per_step", "cpu_baseline", "lba", "pose_opt")})
