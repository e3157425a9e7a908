#!/usr/bin/env python3
"""Hot-path benchmark (driver contract: python bench.py --gpus N --steps K --warmup W).

Workload (BASELINE.json configs[1]+[2], the configuration the ">= 10 000 fps ORB extract+match at 1 GPU" target is quoted
on): synthetic 640x480 frames, 8-level pyramid, 1000 features/frame, on 1 x MI355X.  One *step* = one batch of `--batch`
frames (already resident in HBM) through the whole extractor (pyramid -> per-cell FAST+NMS -> quadtree -> IC_Angle ->
Gaussian blur -> rBRIEF) followed by brute-force 256-bit Hamming matching of every frame against its successor in the
batch; key-points, descriptors and match indices stay in HBM.
With N > 1 (one process per GPU, RCCL) every rank extracts and matches its own `--batch` frames of the rumination queue
(weak scaling, configs[4]) and the step ends with the all-gather of (counts, key-points, descriptors).
The JSON line also carries an `lba` object: BASELINE.json configs[3] (20 key-frames x 3000 map points) on the GPU next to
the CPU oracle, and `pose_opt` (PoseOptimization, 256 frames x 300 correspondences in one launch).

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (per-stage device time from HIP events
on the stream the kernels run on); `cpu_baseline` is the CPU oracle (kind "port": the reference itself cannot
be built in this image) timed single-threaded on a bounded sample of the same frames, rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def algorithmic_bytes_per_frame(n_kp, w=640, h=480):
    """4 752 128 + n*(1369+749+60) B for 640x480, 8 levels (general form below)."""
    lv, ww, hh = [], w, h
    import numpy as np
    inv = np.float32(1.0)
    sc = np.float32(1.0)
    for l in range(8):
        if l:
            sc = np.float32(np.float64(sc) * np.float64(np.float32(1.2)))
        inv = np.float32(1.0) / sc
        lv.append((int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))))
    px = [a * b for a, b in lv]
    total = sum(px)
    return dict(read_l0=px[0], write_levels=total - px[0], fast_read=total, blur_rw=2 * total,
                per_kp=1369 + 749 + 60, total=px[0] + (total - px[0]) + total + 2 * total + n_kp * (1369 + 749 + 60))


def side_legs(args):
    """BASELINE.json configs[3] (local BA) and PoseOptimization, GPU next to the CPU oracle (rank 0, N = 1)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from ba_scene import ba_problem, pose_problem
    from rumi_slam_amd.optimizer import Optimizer
    opt = Optimizer()
    b = ba_problem(seed=0, n_opt=20, n_fixed=5, n_points=3000)
    a = (b["kf_pose"], b["kf_fixed"], b["mp_pos"], b["e_mp"], b["e_kf"], b["e_obs"], b["e_w"], b["K"])
    opt.LocalBundleAdjustment(*a)                                   # warm-up
    g = []
    for _ in range(5):
        t0 = time.perf_counter(); stats, kp, mp, er = opt.LocalBundleAdjustment(*a); g.append(time.perf_counter() - t0)
    dev_ms = float(opt.stage_ms()[5])
    c = []
    for _ in range(3):
        t0 = time.perf_counter(); its, kpr, mpr, err = oracle_lib.local_ba(*a); c.append(time.perf_counter() - t0)
    rel = float(np.max(np.linalg.norm(mp - mpr, axis=1) / np.maximum(np.linalg.norm(mpr, axis=1), 1e-2)))
    lba = {"workload": "LocalBundleAdjustment: 20 optimised + 5 fixed key-frames x 3000 map points (BASELINE.json configs[3])",
           "edges": int(len(b["e_mp"])), "lm_iterations": int(stats[0]), "lm_trials": int(stats[1]),
           "gpu_ms_wall": round(min(g) * 1e3, 3), "gpu_ms_device": round(dev_ms, 3), "cpu_ms": round(min(c) * 1e3, 2),
           "speedup_wall": round(min(c) / min(g), 1), "max_rel_landmark_diff_vs_oracle": rel, "cpu_cores": 1}
    probs = [pose_problem(100 + i, 300, 0.1) for i in range(256)]
    start = np.cumsum([0] + [len(p["inv_sigma2"]) for p in probs]).astype(np.int32)
    pa = (start, np.concatenate([p["Xw"] for p in probs]), np.concatenate([p["obs"] for p in probs]),
          np.concatenate([p["inv_sigma2"] for p in probs]), probs[0]["K"], np.stack([p["T0"] for p in probs]))
    opt.PoseOptimizationBatch(*pa)
    t0 = time.perf_counter(); opt.PoseOptimizationBatch(*pa); gp = time.perf_counter() - t0
    t0 = time.perf_counter()
    for p in probs[:64]:
        oracle_lib.pose_optimization(p["Xw"], p["obs"], p["inv_sigma2"], p["K"], p["T0"])
    cp = (time.perf_counter() - t0) / 64
    pose = {"workload": "PoseOptimization: 256 frames x 300 correspondences, one launch (host arrays in, PCIe included)",
            "gpu_us_per_frame": round(gp / 256 * 1e6, 2), "cpu_us_per_frame": round(cp * 1e6, 2), "speedup": round(cp / (gp / 256), 1)}
    return {"lba": lba, "pose_opt": pose}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="frames per step per GPU (the rumination queue of BASELINE.json configs[4]; launches cover 256 frames)")
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--unique", type=int, default=32, help="distinct synthetic frames (tiled to --batch)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU-oracle baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from rumi_slam_amd.extractor import ORBextractor
    from rumi_slam_amd.matcher import bruteforce_batch
    from rumi_slam_amd import rumination
    from rumi_slam_amd.synth import synth_frame

    B, W, H = args.batch, 640, 480
    uniq = min(args.unique, B)
    host = np.stack([synth_frame(1234 + rank * 100000 + i) for i in range(uniq)])
    frames = torch.from_numpy(host).to(dev)
    frames = frames.repeat((B + uniq - 1) // uniq, 1, 1)[:B].contiguous()

    ext = ORBextractor(args.nfeatures, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B, device=local_rank)
    cap = args.nfeatures + 4 * 8 + 64

    pending = [None]

    def step():
        kp, desc, counts = ext.extract_batch(frames, (0, 1000), cap=cap, wait=False)    # enqueue only: the host queues step i + 1 while step i runs
        # frame i against frame i+1 (the last one against the first): B independent 1000 x 1000 problems
        # (views of the extractor's output: the successor of frame i is the same buffer one record further, no copy; the last pair wraps)
        match = (bruteforce_batch(desc[:-1], counts[:-1], desc[1:], counts[1:]) if B > 1 else None,
                 bruteforce_batch(desc[-1:], counts[-1:], desc[:1], counts[:1]))
        if world > 1:
            # the path's one exchange step: every GPU ends up with all key-points / descriptors.  It is launched
            # here and joined after the NEXT step's kernels are queued (RCCL runs on its own stream), so the exchange of step i
            # overlaps the extraction of step i + 1; drain() joins the last one inside the timed region.
            prev, pending[0] = pending[0], rumination.all_gather_records_async(counts, kp, desc, B * world)
            if prev is not None:
                prev.wait()
        return kp, desc, counts, match

    def drain():
        if pending[0] is not None:
            gc, gk, gd = pending[0].wait()
            pending[0] = None
            return gk, gd, gc
        return None

    for _ in range(args.warmup):
        out = step()
    drain()
    ext.sync()
    torch.cuda.synchronize()
    n_kp = float(out[2].reshape(-1, 2)[:, 0].float().mean().item())

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    drain()
    ext.sync()                                    # waits for the last step and raises on any device-side capacity condition of the K steps
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # per-stage device time of one more (untimed) step, HIP events on the kernels' stream
    ext.set_profiling(True)
    ext.extract_batch(frames, (0, 1000), cap=cap)
    torch.cuda.synchronize()
    stage = {k: float(v) for k, v in ext.stage_ms().items()}
    ext.set_profiling(False)

    if rank == 0:
        fps = B * world * args.steps / dt
        ab = algorithmic_bytes_per_frame(n_kp)
        # dominant kernel by device time; algorithmic bytes of that kernel per launch
        kern_bytes = {"fast": ab["fast_read"], "pyramid": ab["read_l0"] + 1.44 * ab["write_levels"] + ab["write_levels"],
                      "blur": ab["blur_rw"], "orient_desc": n_kp * ab["per_kp"], "quadtree": 0.0}
        kern_ms = {k: stage[k] for k in kern_bytes}
        dom = max(kern_ms, key=kern_ms.get)
        # the profiled pass runs the batch in launches of up to 256 frames on one stream; stage times are sums over those launches
        per_launch = min(B, 256)
        n_launch = (B + 255) // 256
        achieved = kern_bytes[dom] * B / (kern_ms[dom] * 1e-3) / 1e9 if kern_ms[dom] > 0 else 0.0
        # HBM bytes per launch from the PMC counters: rocprofv3 cannot wrap this process from inside, so the number is the
        # committed result of `tools/pmc_summary.py` on two --pmc passes of THIS command line (profiles/r01_pmc_traffic.json);
        # it is used only when it was taken at the same frames-per-launch.
        traffic = None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            kname = {"fast": "k_fast_cells", "blur": "k_blur", "orient_desc": "k_orient_desc", "quadtree": "k_octree", "pyramid": "k_resize"}[dom]
            if pm.get("frames_per_launch") == per_launch and kname in pm["kernels"]:
                traffic = pm["kernels"][kname]["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
        # VALU issue accounting of the same kernel: wave-instructions per launch from the committed SQ_INSTS_VALU pass
        # (profiles/r01_pmc_sq_counters.json) over the chip's measured integer issue rate (tools/valu_rate.hip: 4.2 cycles per wave64
        # instruction and SIMD = 585 G wave-instructions/s) and this run's launch duration
        valu = None
        try:
            sq = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_sq_counters.json")))["kernels"]
            insts = sq[{"fast": "k_fast_cells", "blur": "k_blur", "orient_desc": "k_orient_desc", "quadtree": "k_octree", "pyramid": "k_resize"}[dom]]["SQ_INSTS_VALU"]
            if per_launch == 256 and kern_ms[dom] > 0:
                valu = {"wave_insts_per_launch": int(insts), "issue_rate_G_per_s": 585.0,
                        "issue_frac": round(insts / 585e9 / (kern_ms[dom] / n_launch * 1e-3), 4)}
        except Exception:
            valu = None
        line = {
            "metric": "frames/sec ORB extract+match", "value": round(fps, 1), "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "ORB extract (640x480, 8-level pyramid, %d features/frame) + brute-force 256-bit Hamming match of consecutive frames (BASELINE.json configs[1]+[2])" % args.nfeatures,
                       "frames_per_step_per_gpu": B, "mean_keypoints_per_frame": round(n_kp, 1),
                       "exchange": "all_gather(counts,keypoints,descriptors) over RCCL" if world > 1 else "none"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(kern_bytes[dom] * per_launch), "frames_per_launch": per_launch,
                         "launch_ms": round(kern_ms[dom] / n_launch, 4),
                         "whole_path_GBps": round(ab["total"] * fps / 1e9, 2),
                         "whole_path_frac": round(ab["total"] * fps / 1e9 / HBM_PEAK_GBS, 5), "valu": valu},
            "stage_ms_per_step": {k: round(v, 3) for k, v in stage.items()},
            "stage_ms_note": "one extra profiled step, every kernel alone on one stream (no overlap), summed over the step's launches",
        }
        if world == 1 and not args.no_cpu:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib                      # the checker, timed as the CPU baseline (kind "port")
            orc = oracle_lib.OracleExtractor(args.nfeatures, 1.2, 8, 20, 7)
            n, c0, prev = 0, time.perf_counter(), None
            while n < len(host) * 8 and time.perf_counter() - c0 < args.cpu_seconds:
                _, _, d = orc.extract(host[n % len(host)], (0, 1000))
                if prev is not None:
                    oracle_lib.bruteforce_match(prev, d)
                prev = d
                n += 1
            cdt = time.perf_counter() - c0
            line["cpu_baseline"] = {"value": round(n / cdt, 2), "unit": "frames/s", "cores": 1, "kind": "port",
                                    "sample": "%d of the same synthetic frames (extract + brute-force match), oracle/ (g++ -O2, scalar, 1 thread), %.1f s" % (n, cdt)}
            line.update(side_legs(args))
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
