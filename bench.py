#!/usr/bin/env python3
"""Hot-path benchmark (driver contract: python bench.py --gpus N --steps K --warmup W).

Workload (BASELINE.json configs[1]+[2], the configuration the ">= 10 000 fps ORB extract+match at 1 GPU" target is quoted
on): synthetic 640x480 frames, 8-level pyramid, 1000 features/frame, on 1 x MI355X.  One *step* = one batch of `--batch`
DISTINCT frames (already resident in HBM) through the whole extractor (pyramid -> per-cell FAST+NMS -> quadtree -> IC_Angle ->
Gaussian blur -> rBRIEF) followed by brute-force 256-bit Hamming matching of every frame against its successor in the
batch; key-points, descriptors and match indices stay in HBM.  Steps are enqueued without waiting for each other
(rumi_orb_extract_batch_device_async); the timed region ends with rumi_orb_sync + a device synchronisation.

N > 1 (one process per GPU, RCCL; `python bench.py --gpus N` starts the N ranks itself when it was not started by torchrun):
BASELINE.json configs[4], the rumination queue of `--batch` frames sharded over the ranks in contiguous time-ordered blocks
(`--scaling strong`, the default: 1024 frames in all, 1024 / N per rank) or `--batch` frames PER rank (`--scaling weak`); each rank
writes fixed-capacity per-frame records and the step ends with the queue's ONE all-gather of them.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (per-stage device time from HIP events on the stream the
kernels run on, every kernel alone on it); `cpu_baseline` is the CPU oracle (kind "port": the reference itself cannot be built in
this image) timed single-threaded on a bounded sample of the same frames, rank 0, N = 1 only.  N = 1 also reports
`value_h2d_inclusive` (the same step with the frames starting in pinned HOST memory, transfers overlapped with the kernels),
`single_frame_host_api_fps` (and `single_frame_c_call_fps`: the C entry alone on the pinned capture buffer), a `batch_sweep`, and the `lba` / `pose_opt` side legs.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
F64_MFMA_PEAK_TFLOPS = 78.6    # dense f64 matrix peak


def algorithmic_bytes_per_frame(n_kp, w=640, h=480):
    """4 752 128 + n*(1369+749+60) B for 640x480, 8 levels (general form below)."""
    lv = []
    import numpy as np
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


def whole_path_valu_issue(fps):
    """The step against the vector-issue ceiling, as ONE number: every kernel's VALU wave-instructions per 256-frame pass (committed SQ counters)
    priced with the measured issue cost of ITS OWN instruction mix -- each opcode of the disassembled hot loops in the 2.4-, 4.2- or 8.2-cycle
    class of profiles/r02_valu_issue_rates.txt, loops weighted by their trip counts (tools/valu_mix.py -> profiles/r04_valu_mix.json) -- over the
    SIMD-cycles available at the rate measured in THIS run."""
    try:
        mix = json.load(open(os.path.join(ROOT, "profiles", "r04_valu_mix.json")))
        simd_cycles = 1024 * 2.4e9 * (256.0 / fps)
        return {"valu_wave_insts_per_256_frames": mix["valu_wave_insts_per_256_frames"], "weighted_cycles": mix["weighted_cycles_per_256_frames"],
                "avg_cycles_per_inst": mix["avg_cycles_per_inst"], "frac_of_weighted_ceiling": round(mix["weighted_cycles_per_256_frames"] / simd_cycles, 4),
                "per_kernel_mix": {k: {"M_insts": round(v["valu_wave_insts_per_256_frames"] / 1e6, 1), "fast": v["fast_frac"], "slow": v["slow_frac"], "v8": v["v8_frac"],
                                       "avg_cycles": v["avg_cycles_per_inst"]} for k, v in mix["kernels"].items()},
                "source": "instruction counts and mix: committed profile profiles/r04_valu_mix.json (separate --pmc passes + disassembly), not this run; rate: this run"}
    except Exception:
        return None


def lba_roofline(dev_ms, trials):
    """MFMA roofline of the local BA's dominant kernel (k_baw_system: linearisation + the Schur product Y^T Y on the f64 matrix cores), from the
    committed counter passes; `this_run` prices the same counted flops against this run's whole device time."""
    out = {"bound": "mfma", "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s"}
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "r04_lba_pmc_mfma.json")))
        for tag in ("batch16", "single"):
            k = prof[tag]["kernels"]["k_baw_system"]
            out[tag] = {"kernel": "k_baw_system", "mfma_flops_per_launch": k["mfma_flops"], "duration_us": k["duration_us"],
                        "achieved": round(k["mfma_flops"] / (k["duration_us"] * 1e-6) / 1e12, 2), "frac": k["mfma_util_vs_78.6TF"],
                        "mfma_busy_cycles": k["SQ_VALU_MFMA_BUSY_CYCLES"], "mfma_instructions": k["SQ_INSTS_VALU_MFMA_F64"]}
        one = prof["single"]["kernels"]
        flops_trial = one["k_baw_system"]["mfma_flops"] + one["k_baw_reduce"]["mfma_flops"] + one["k_baw_solve"]["mfma_flops"]
        out["this_run"] = {"device_ms": round(dev_ms, 3), "lm_trials": trials, "mfma_flops_per_trial_counted": flops_trial,
                           "frac_of_whole_device_time": round(flops_trial * trials / (dev_ms * 1e-3) / 1e12 / F64_MFMA_PEAK_TFLOPS, 4)}
        out["source"] = "profiles/r04_lba_pmc_mfma.json: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_F64 (tools/pmc_lba.sh), not this run; this_run: HIP events of this run"
    except Exception as e:
        out["error"] = str(e)
    return out


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
    st = opt.stage_ms()
    dev_ms = float(st[5])
    c = []
    for _ in range(3):
        t0 = time.perf_counter(); its, kpr, mpr, err = oracle_lib.local_ba(*a); c.append(time.perf_counter() - t0)
    rel = float(np.max(np.linalg.norm(mp - mpr, axis=1) / np.maximum(np.linalg.norm(mpr, axis=1), 1e-2)))
    E, K, trials = int(len(b["e_mp"])), 20, max(int(stats[1]), 1)
    lba = {"workload": "LocalBundleAdjustment: 20 optimised + 5 fixed key-frames x 3000 map points (BASELINE.json configs[3])",
           "edges": E, "lm_iterations": int(stats[0]), "lm_trials": int(stats[1]),
           "gpu_ms_wall": round(min(g) * 1e3, 3), "gpu_ms_device": round(dev_ms, 3), "cpu_ms": round(min(c) * 1e3, 2),
           "speedup_wall": round(min(c) / min(g), 1), "max_rel_landmark_diff_vs_oracle": rel, "cpu_cores": 1}
    lba["gpu_ms_c_call"] = round(opt.last_call_s * 1e3, 3)
    # MFMA accounting from the matrix-core COUNTERS of the committed rocprofv3 passes (tools/pmc_lba.sh -> profiles/r04_lba_pmc_mfma.json: the
    # launches of this very workload, 16 windows per launch and one), next to this run's own device time
    lba["roofline"] = lba_roofline(dev_ms, trials)
    # R independent windows: the window is a batch dimension of the kernels, ONE host thread drives the batch (rumi_local_ba_batch)
    try:
        R = 16
        opt.LocalBundleAdjustmentBatch([a] * R, 1)
        bts, cpu = [], []
        for _ in range(5):
            c0 = time.process_time(); opt.LocalBundleAdjustmentBatch([a] * R, 1); cpu.append(time.process_time() - c0); bts.append(opt.last_call_s)
        lba["batch"] = {"windows": R, "host_threads": 1, "ms_per_window": round(min(bts) / R * 1e3, 4), "ms_per_window_median": round(sorted(bts)[2] / R * 1e3, 4),
                        "host_cores_busy": round(sorted(cpu)[2] / (sorted(bts)[2] + 1e-9), 2),
                        "note": "the C entry alone (the Python mirror's per-window array copies excluded); host_cores_busy = process CPU time of the whole Python call / wall time of the C call (an upper bound)"}
    except Exception as e:
        lba["batch"] = {"error": str(e)}
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
    # one Tracking-thread frame (BASELINE.json configs[0]'s path on a synthetic plane scene): the fused device-resident entry next to the separate ones
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import track_probe
        track = track_probe.measure(40)
    except Exception as e:
        track = {"error": str(e)}
    return {"lba": lba, "pose_opt": pose, "tracking_frame": track}


def spawn_ranks(args):
    """`python bench.py --gpus N` without torchrun: start the N ranks as fresh child processes (the parent never touches the GPU) and
    pass rank 0's line through."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    raise SystemExit(rc)


def make_frames(torch, np, dev, n, rank):
    """n DISTINCT 640x480 frames on the device: 32 seeded synthetic frames (SURVEY.md section 8d generator, host) x cyclic shifts of them
    (device), so that no two frames of a step share their pixels (host synthesis of 1024 frames would take minutes)."""
    from rumi_slam_amd.synth import synth_frame
    nb = min(32, n)
    host = np.stack([synth_frame(1234 + rank * 100000 + i) for i in range(nb)])
    base = torch.from_numpy(host).to(dev)
    fr = torch.empty((n, 480, 640), dtype=torch.uint8, device=dev)
    for k in range(n):
        fr[k] = torch.roll(base[k % nb], shifts=(7 * (k // nb), 11 * (k // nb)), dims=(0, 1)) if k >= nb else base[k]
    return fr, host


def one_process_queue(args):
    """`bench.py --gpus N --one-process`: the reference is ONE C++ process (System.cc:193-240), so is this leg: rumi_queue_extract cuts the host
    queue into N contiguous blocks, one extractor per device, ONE ncclAllGather of the records (include/rumi_queue.h).  Extraction only."""
    import numpy as np
    sys.path.insert(0, ROOT)
    from rumi_slam_amd.queue import RuminationQueue
    from rumi_slam_amd.synth import synth_frame
    N, F = args.gpus, args.batch
    logical = bool(os.environ.get("RUMI_BENCH_LOGICAL_SHARDS"))
    base = [synth_frame(1234 + i) for i in range(32)]
    frames = [base[i] if i < 32 else np.roll(base[i % 32], (7 * (i // 32), 11 * (i // 32)), (0, 1)) for i in range(F)]
    q = RuminationQueue(args.nfeatures, 1.2, 8, 20, 7, [0] * N if logical else list(range(N)), max_block=(F + N - 1) // N, cap=args.nfeatures + 96)
    import torch
    rec = torch.zeros((F, q.record_bytes), dtype=torch.uint8).pin_memory().numpy()
    for _ in range(args.warmup):
        q.extract(frames, (0, 1000), out=rec)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        q.extract(frames, (0, 1000), out=rec)
    dt = time.perf_counter() - t0
    emit(dict({"metric": "frames/sec ORB extract, rumination queue from one process (host frames in, gathered records on every device, host records out)",
                      "value": round(F * args.steps / dt, 1), "unit": "frames/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
                      "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8",
                      "data": "synthetic", "config": {"workload": "BASELINE.json configs[4]: %d queued 640x480 frames over %d %s, one process, rumi_queue_extract" % (F, N, "logical shards on device 0" if logical else "devices"),
                                                      "exchange": "RCCL ncclAllGather" if q.uses_rccl else "device-to-device copies (logical shards)"},
                      "last_call_ms": q.last_ms()}))


_LINE_OUT = None


def claim_stdout():
    """stdout carries the ONE JSON line and nothing else: libraries that greet on stdout (RCCL prints a version banner at communicator creation on
    this image) are sent to stderr, the line goes to the original descriptor."""
    global _LINE_OUT
    if _LINE_OUT is None:
        sys.stdout.flush()
        _LINE_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit(line):
    out = _LINE_OUT or sys.stdout
    out.write(json.dumps(line) + "\n"); out.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="frames of the rumination queue per step (BASELINE.json configs[4]); per GPU with --scaling weak")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong", help="N > 1: --batch frames in all (strong) or per rank (weak)")
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU-oracle baseline leg")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline and the side legs (profiling runs)")
    ap.add_argument("--one-process", action="store_true", help="the queue over --gpus devices from ONE process through include/rumi_queue.h (RCCL all-gather inside the library) instead of one rank per GPU; "
                    "RUMI_BENCH_LOGICAL_SHARDS=1 aliases every shard to device 0 (a one-GPU box)")
    args = ap.parse_args()
    if args.one_process:
        claim_stdout()
        return one_process_queue(args)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    claim_stdout()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from rumi_slam_amd.extractor import ORBextractor
    from rumi_slam_amd.matcher import bruteforce_ring
    from rumi_slam_amd import rumination

    W, H = 640, 480
    if world > 1 and args.scaling == "strong":
        n_queue = args.batch                                   # the whole queue; this rank owns a contiguous block of it
        lo, hi = rumination.shard_bounds(n_queue, rank, world)
        B = hi - lo
    else:
        n_queue, B = args.batch * world, args.batch
    per = rumination.shard_capacity(n_queue, world)            # records per rank in the all-gather (blocks padded to it)
    frames, host = make_frames(torch, np, dev, B, rank)

    ext = ORBextractor(args.nfeatures, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B, device=local_rank)
    cap = args.nfeatures + 4 * 8 + 64

    xchg_stream = torch.cuda.Stream(dev)
    ext.set_resident_queue(True)          # the queue sits in HBM before the timed region: no call waits for the stream it is issued on (include/rumi_orb.h)

    def match_pairs(desc, counts, out=None):
        # frame i against frame i+1 (the last one against the first): independent 1000 x 1000 problems, one launch over the extractor's output
        # in place (the successor of frame i is the same buffer one record further)
        return bruteforce_ring(desc, counts, out=out)

    class Step:
        """One step over `fr` frames.  Outputs live in FOUR preallocated buffer sets used in turn (nothing is allocated or cleared inside the
        timed region); a set is handed to the extractor again only behind the event that marks the end of its previous consumers (matching,
        exchange): rumi_orb_wait_event.  records=True is the N > 1 code path: per-frame records written in place + the queue's one all-gather."""

        def __init__(self, fr, records, n_queue_=None, per_=None, slots=None):
            n = fr.shape[0]
            if slots is None:                                     # (more than four calls in flight measured no faster at any queue length)
                slots = int(os.environ.get("RUMI_BENCH_SLOTS", "0")) or 4
            ext.set_resident_queue(slots)
            self.fr, self.records, self.i, self.nbuf = fr, records, 0, slots  # as many buffer sets as the extractor has slots: that many calls in flight
            self.n_queue = n_queue_ if n_queue_ is not None else n
            if records:
                self.per = per_ if per_ is not None else n
                self.rec = [torch.zeros((self.per, rumination.record_bytes(cap)), dtype=torch.uint8, device=dev) for _ in range(self.nbuf)]
                self.views = [rumination.record_views(r[:n], cap) for r in self.rec]
                self.gbuf = [torch.empty((world * self.per, rumination.record_bytes(cap)), dtype=torch.uint8, device=dev) if world > 1 else None for _ in range(self.nbuf)]
            else:
                self.out = [(torch.empty((n, cap, 7), dtype=torch.float32, device=dev), torch.empty((n, cap, 32), dtype=torch.uint8, device=dev),
                             torch.zeros((n, 2), dtype=torch.int32, device=dev)) for _ in range(self.nbuf)]
            self.mout = [[torch.empty((n, cap), dtype=torch.int32, device=dev) for _ in range(3)] for _ in range(self.nbuf)]
            self.consumed = [None] * self.nbuf
            self.ev_match = [torch.cuda.Event() for _ in range(self.nbuf)]       # (events are made once: creating two per step shows at 64 frames per step)
            self.ev_join = [torch.cuda.Event() for _ in range(self.nbuf)]
            self.gather = [None] * self.nbuf
            torch.cuda.synchronize()

        def __call__(self):
            k = self.i % self.nbuf
            self.i += 1
            if self.gather[k] is not None:                       # the exchange that still reads this set: its end and the matching's, as ONE event
                if self.gather[k]._work is not None:             # (a stream that carries nothing else: no false dependency on later steps)
                    with torch.cuda.stream(xchg_stream):
                        self.gather[k].wait()
                xchg_stream.wait_event(self.consumed[k])
                self.ev_join[k].record(xchg_stream); self.consumed[k] = self.ev_join[k]
                self.gather[k] = None
            if self.consumed[k] is not None:
                ext.wait_event(self.consumed[k])
            if self.records:
                ext.extract_batch_records(self.fr, (0, 1000), cap=cap, wait=False, out=self.rec[k])
                kp, desc, counts = self.views[k]
            else:
                kp, desc, counts = ext.extract_batch(self.fr, (0, 1000), cap=cap, wait=False, out=self.out[k])   # enqueue only
            # The matching of step i follows on the caller's stream, which the call above has made wait for the extraction of step i; the
            # resident extractor never waits for this stream, so the wide Hamming kernel shares the device with the latency-bound stretches
            # (quadtree, compaction, upper pyramid levels) of the next steps without a stream of its own (a separate one measured 1-13 % slower).
            if self.records:
                # the path's one exchange step: every GPU ends up with all records.  Launched behind the extraction only
                # (RCCL runs on its own stream) and joined when its buffer set comes round again, so it overlaps the next steps' kernels
                self.gather[k] = rumination.all_gather_records_async(self.rec[k], self.n_queue, out=self.gbuf[k])
            m = match_pairs(desc, counts, self.mout[k])
            self.ev_match[k].record(); self.consumed[k] = self.ev_match[k]
            return kp, desc, counts, m

        def drain(self):
            for k in range(self.nbuf):
                if self.gather[k] is not None:
                    self.gather[k].wait()
                    self.gather[k] = None

    use_records = world > 1 or bool(os.environ.get("RUMI_BENCH_FORCE_RECORDS"))   # (the env switch runs the N > 1 code path on one GPU: its exchange degenerates to a no-op)
    step = Step(frames, use_records, n_queue, per)

    def drain():
        step.drain()

    def timed(fn, steps, warmup):
        drain_ = fn.drain if hasattr(fn, "drain") else (lambda: None)
        for _ in range(warmup):
            out = fn()
        drain_(); ext.sync(); torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = fn()
        drain_()
        ext.sync()                                # waits for the last step and raises on any device-side capacity condition of the steps
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, out

    dt, out = timed(step, args.steps, args.warmup)
    n_kp = float(out[2][:, 0].float().mean().item())

    # per-stage device time of one more (untimed) pass with every kernel ALONE on one stream in launches of up to 256 frames
    # (rumi_orb_set_profiling: HIP events recorded by the library on the stream the kernels run on)
    ext.set_resident_queue(False)
    ext.extract_batch(frames, (0, 1000), cap=cap)           # (this mode's arenas and code paths warm)
    torch.cuda.synchronize()
    ext.set_profiling(True)
    passes = []
    for _ in range(5):
        ext.extract_batch(frames, (0, 1000), cap=cap)
        torch.cuda.synchronize()
        passes.append({k: float(v) for k, v in ext.stage_ms().items()})
    stage = {k: sorted(p[k] for p in passes)[len(passes) // 2] for k in passes[0]}      # per stage: the median of five passes
    ext.set_profiling(False)
    ext.set_resident_queue(True)

    if rank == 0:
        fps = n_queue * args.steps / dt
        ab = algorithmic_bytes_per_frame(n_kp)
        # dominant kernel by device time; algorithmic bytes of that kernel per launch
        # (pyramid: SURVEY.md section 8d's figure -- level 0 read once, levels 1-7 written once; the re-reads of levels 1-6 as sources are traffic, not algorithm)
        kern_bytes = {"fast": ab["fast_read"], "pyramid": ab["read_l0"] + ab["write_levels"],
                      "blur": ab["blur_rw"], "orient_desc": n_kp * ab["per_kp"], "quadtree": 0.0}
        kern_ms = {k: stage[k] for k in kern_bytes}
        dom = max(kern_ms, key=kern_ms.get)
        per_launch = min(B, 256)
        n_launch = (B + 255) // 256
        achieved = kern_bytes[dom] * B / (kern_ms[dom] * 1e-3) / 1e9 if kern_ms[dom] > 0 else 0.0
        kname = {"fast": "k_fast_cells", "blur": "k_blur", "orient_desc": "k_orient_desc", "quadtree": "k_octree", "pyramid": "k_resize"}[dom]
        # HBM bytes per launch from the PMC counters: rocprofv3 cannot wrap this process from inside, so these two numbers are the committed
        # results of separate --pmc passes of this command line (profiles/r04_pmc_*.json); `*_source` says so, and they are used only when
        # taken at the same frames-per-launch.
        traffic, valu = None, None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_traffic.json")))
            if pm.get("frames_per_launch") == per_launch and kname in pm["kernels"]:
                traffic = pm["kernels"][kname]["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
        try:
            sq = json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_sq_counters.json")))["kernels"][kname]
            if per_launch == 256:
                valu = {"wave_insts_per_launch": int(sq["SQ_INSTS_VALU"]), "lds_insts_per_launch": int(sq.get("SQ_INSTS_LDS", 0)),
                        "lds_bank_conflict_cycles": int(sq.get("SQ_LDS_BANK_CONFLICT", 0)),
                        "note": "instruction classes issue at 2.4 (add/sub/logic/shift-right) or 4.2 (min/max, 3-operand, packed) cycles per wave and SIMD: profiles/r02_valu_issue_rates.txt"}
        except Exception:
            valu = None
        line = {
            "metric": "frames/sec ORB extract+match", "value": round(fps, 1), "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": args.scaling if world > 1 else "n/a", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic (32 seeded frames per rank x cyclic shifts: every frame of a step distinct)",
            "config": {"workload": "ORB extract (640x480, 8-level pyramid, %d features/frame) + brute-force 256-bit Hamming match of consecutive frames (BASELINE.json configs[1]+[2]%s)" % (args.nfeatures, "; queue sharded as configs[4]" if world > 1 else ""),
                       "frames_per_step": n_queue, "frames_per_step_per_gpu": B, "scaling": args.scaling if world > 1 else "n/a (one GPU)",
                       "mean_keypoints_per_frame": round(n_kp, 1),
                       "exchange": "one all_gather_into_tensor of %d-byte per-frame records over RCCL, overlapped with the next step" % rumination.record_bytes(cap) if world > 1 else "none"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "traffic_source": "committed profile profiles/r04_pmc_traffic.json (separate --pmc passes), not this run" if traffic else None,
                         "algorithmic_bytes_per_launch": int(kern_bytes[dom] * per_launch), "frames_per_launch": per_launch,
                         "launch_ms": round(kern_ms[dom] / n_launch, 4),
                         "whole_path_GBps": round(ab["total"] * fps / 1e9, 2),
                         "whole_path_frac": round(ab["total"] * fps / 1e9 / HBM_PEAK_GBS, 5), "valu": valu,
                         "valu_source": "committed profile profiles/r04_pmc_sq_counters.json, not this run" if valu else None},
            "valu_issue": whole_path_valu_issue(fps),
            "stage_ms_per_step": {k: round(v, 3) for k, v in stage.items()},
            "stage_ms_note": "five extra profiled steps (per stage the median): every kernel alone on ONE stream (RUMI_SERIAL-equivalent: the blur too), launches of up to 256 frames, summed over the step's launches",
        }
        if world == 1 and not args.no_cpu:
            # ---- the same step with the frames starting in pinned HOST memory (the queue as the reference holds it), transfers overlapped ----
            hostq = frames.cpu().pin_memory()

            def step_h2d():
                kp, desc, counts = ext.extract_batch_host(hostq, (0, 1000), cap=cap)
                return kp, desc, counts, match_pairs(desc, counts)
            dth, _ = timed(step_h2d, max(3, args.steps // 2), 1)
            line["value_h2d_inclusive"] = round(B * max(3, args.steps // 2) / dth, 1)
            line["h2d_note"] = "%d x 307 200 B per step from pinned host memory, 64-frame groups on a copy stream under the kernels (rumi_orb_extract_batch_host)" % B
            # ---- ... and with the results brought back to the host as well: key-points, descriptors, counts and the match indices land in pinned host
            # buffers (preallocated: nothing is allocated in the timed region).  Host frames in, host results out: what a host consumer of the queue sees.
            import ctypes as C_
            from rumi_slam_amd import capi as capi_
            hk = torch.empty((B, cap, 7), dtype=torch.float32).pin_memory(); hd = torch.empty((B, cap, 32), dtype=torch.uint8).pin_memory()
            hc = torch.zeros((B, 2), dtype=torch.int32).pin_memory()
            hm = [torch.empty((B, cap), dtype=torch.int32).pin_memory() for _ in range(3)]
            okp = torch.empty((B, cap, 7), dtype=torch.float32, device=dev); odesc = torch.empty((B, cap, 32), dtype=torch.uint8, device=dev)
            ocnt = torch.zeros((B, 2), dtype=torch.int32, device=dev)
            mo = [torch.empty((B, cap), dtype=torch.int32, device=dev) for _ in range(3)]
            ptrs = (C_.c_void_p * B)(*[hostq.data_ptr() + f * hostq.stride(0) for f in range(B)])

            def step_h2d_d2h():
                st_ = torch.cuda.current_stream(dev)
                capi_.check(ext._lib.rumi_orb_extract_batch_host(ext._h, ptrs, B, W, H, hostq.stride(1), 0, 1000, okp.data_ptr(), odesc.data_ptr(), ocnt.data_ptr(), cap,
                                                                 hk.data_ptr(), hd.data_ptr(), hc.data_ptr(), st_.cuda_stream))
                m = match_pairs(odesc, ocnt, mo)
                for a_, b_ in zip(hm, m):
                    a_.copy_(b_, non_blocking=True)
                torch.cuda.synchronize()
                return okp, odesc, ocnt, m
            dtd, _ = timed(step_h2d_d2h, max(3, args.steps // 2), 5)      # (fresh pinned buffers are slow on their first use by each stream that copies into them)
            line["value_h2d_d2h_inclusive"] = round(B * max(3, args.steps // 2) / dtd, 1)
            line["h2d_d2h_note"] = "host frames in (pinned), key-points + descriptors + counts + match indices out to pinned host buffers (%.1f MB per step back); every step ends synchronised" % ((hk.numel() * 4 + hd.numel() + hc.numel() * 4 + 3 * hm[0].numel() * 4) / 1e6)
            # ---- the same with the way back of step i under the uploads of step i + 1: two sets of device and pinned host buffers, the copies of a step on a
            # stream of their own behind an event; a host consumer gets step i's results one step later
            cb = torch.cuda.Stream(dev)
            sets = [dict(okp=okp, odesc=odesc, ocnt=ocnt, mo=mo, hk=hk, hd=hd, hc=hc, hm=hm, done=torch.cuda.Event())]
            sets.append(dict(okp=torch.empty_like(okp), odesc=torch.empty_like(odesc), ocnt=torch.zeros_like(ocnt), mo=[torch.empty_like(x) for x in mo],
                             hk=torch.empty_like(hk).pin_memory(), hd=torch.empty_like(hd).pin_memory(), hc=torch.zeros_like(hc).pin_memory(),
                             hm=[torch.empty_like(x).pin_memory() for x in hm], done=torch.cuda.Event()))
            turn = [0]

            def step_pipelined():
                S = sets[turn[0] & 1]; turn[0] += 1
                S["done"].synchronize()                      # the set's previous results have left (the consumer is done with the pinned buffers by then)
                st_ = torch.cuda.current_stream(dev)
                capi_.check(ext._lib.rumi_orb_extract_batch_host(ext._h, ptrs, B, W, H, hostq.stride(1), 0, 1000, S["okp"].data_ptr(), S["odesc"].data_ptr(), S["ocnt"].data_ptr(), cap,
                                                                 None, None, None, st_.cuda_stream))
                m = match_pairs(S["odesc"], S["ocnt"], S["mo"])
                ready = torch.cuda.Event(); ready.record(st_)
                with torch.cuda.stream(cb):
                    cb.wait_event(ready)
                    S["hk"].copy_(S["okp"], non_blocking=True); S["hd"].copy_(S["odesc"], non_blocking=True); S["hc"].copy_(S["ocnt"], non_blocking=True)
                    for a_, b_ in zip(S["hm"], m):
                        a_.copy_(b_, non_blocking=True)
                    S["done"].record(cb)
                return S["okp"], S["odesc"], S["ocnt"], m
            dtp, _ = timed(step_pipelined, max(4, args.steps // 2), 6)
            line["value_h2d_d2h_inclusive_pipelined"] = round(B * max(4, args.steps // 2) / dtp, 1)
            line["h2d_d2h_pipelined_note"] = "the same bytes both ways, the copies back of step i on their own stream under the uploads of step i + 1 (two buffer sets; results one step late)"
            del sets
            del hk, hd, hm, okp, odesc, mo
            # ---- one frame at a time through the drop-in host API (ORBextractor::operator(): host image in, host key-points out) ----
            ext1 = ORBextractor(args.nfeatures, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=1, device=local_rank)
            for i in range(8):
                ext1(host[i % len(host)], None, (0, 1000))
            t0 = time.perf_counter()
            for i in range(200):
                ext1(host[i % len(host)], None, (0, 1000))
            line["single_frame_host_api_fps"] = round(200 / (time.perf_counter() - t0), 1)
            # (the same with the frame captured into the handle's pinned staging memory, rumi_orb_image_buffer, and the C entry timed alone:
            # the Python mirror above allocates and copies the result arrays of every call)
            pin = ext1.image_buffer(W, H)
            c_real, c_spent = ext1._lib.rumi_orb_extract, []
            def c_timed(*a):
                t1 = time.perf_counter(); rc = c_real(*a); c_spent.append(time.perf_counter() - t1); return rc
            ext1._lib.rumi_orb_extract = c_timed
            try:
                for i in range(208):
                    pin[:] = host[i % len(host)]
                    ext1(pin, None, (0, 1000))
            finally:
                ext1._lib.rumi_orb_extract = c_real
            line["single_frame_c_call_fps"] = round(1.0 / float(np.median(c_spent[8:])), 1)
            ext1.close()
            # ---- frames per call: device-resident extract + match, the step above at other queue lengths (a rank's share of configs[4] is 128) ----
            sweep, recs = {}, {}
            for nb in (64, 128, 256, 1024):
                if nb > B:
                    continue
                reps = max(12, 16384 // nb)
                dts, _ = timed(Step(frames[:nb], False), reps, 6)
                sweep[str(nb)] = round(nb * reps / dts, 1)
                if nb in (128, 1024):
                    dts, _ = timed(Step(frames[:nb], True), reps, 6)        # the N > 1 step on this one GPU (its all-gather degenerates to a no-op)
                    recs[str(nb)] = round(nb * reps / dts, 1)
            line["batch_sweep_fps"] = sweep
            line["records_path_fps"] = recs
            line["records_path_note"] = "the step of the N > 1 code path (per-frame records written in place, exchange launched and joined) run on ONE GPU, where the all-gather moves nothing: what a rank of configs[4] does between collectives"
            # ---- other inputs of SURVEY.md section 8d through the same step (256 frames): the headline frames are far denser in corners than images ----
            from rumi_slam_amd.synth import synth_frame
            inputs = {"sparse (40 rectangles: the corner density of natural images)": dict(n_rect=40),
                      "low texture (60 rectangles of contrast 8-19: most cells take the minThFAST retry)": dict(n_rect=60, contrast=(8, 19))}
            isweep = {}
            for name, kw in inputs.items():
                hs = np.stack([synth_frame(5000 + i, **kw) for i in range(16)])
                bs = torch.from_numpy(hs).to(dev)
                fr = torch.empty((256, H, W), dtype=torch.uint8, device=dev)
                for k in range(256):
                    fr[k] = torch.roll(bs[k % 16], shifts=(7 * (k // 16), 11 * (k // 16)), dims=(0, 1)) if k >= 16 else bs[k]
                st_ = Step(fr, False)
                dts, o = timed(st_, 12, 3)
                isweep[name] = {"fps": round(256 * 12 / dts, 1), "mean_keypoints": round(float(o[2][:, 0].float().mean().item()), 1)}
            dts, _ = timed(Step(frames[:256], False), 12, 3)
            isweep["headline frames (400 rectangles), same 256-frame step"] = {"fps": round(256 * 12 / dts, 1)}
            line["input_sweep_fps"] = isweep
            # ---- CPU baseline ----
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
        if world == 1 and not args.no_cpu and hostq is not None:
            # (last of the N = 1 legs: RCCL's communicator leaves helper threads behind, and the one-frame host path measured after it ran at 2.8 k fps instead of 8 k)
            # ---- the queue behind the C ABI, ONE process (include/rumi_queue.h): host frames in, every shard's device holds the gathered records, host
            # records out.  One shard = this GPU (its exchange is RCCL's ncclAllGather on one rank); two LOGICAL shards on this GPU for the sharded code path.
            try:
                from rumi_slam_amd.queue import RuminationQueue
                ql = {}
                nq = min(B, 512)
                hq = [hostq[f].numpy() for f in range(nq)]
                for shards in (1, 2):
                    rq = RuminationQueue(args.nfeatures, 1.2, 8, 20, 7, [local_rank] * shards, max_block=(nq + shards - 1) // shards, cap=cap)
                    rec = torch.zeros((nq, rq.record_bytes), dtype=torch.uint8).pin_memory().numpy()       # (pinned: the records arrive sub-chunk by sub-chunk under the kernels)
                    for _ in range(4):                          # (a fresh pinned buffer is slow on its first use by each of the extractor's streams: 10 ms instead of 3.8)
                        rq.extract(hq, (0, 1000), out=rec)
                    t0 = time.perf_counter()
                    for _ in range(3):
                        rq.extract(hq, (0, 1000), out=rec)
                    ql["%d_shard%s" % (shards, "s_logical" if shards > 1 else "")] = {"fps": round(3 * nq / (time.perf_counter() - t0), 1), "exchange": "RCCL all-gather" if rq.uses_rccl else "device-to-device copies",
                                                                                     "last_ms": {k: round(v, 3) for k, v in rq.last_ms().items()}}
                    rq.close()
                line["queue_c_abi_one_process"] = dict(frames=nq, note="rumi_queue_extract: extraction only (no matching), pinned host frames in, host records out to pinned memory (every shard from its own device, under the kernels)", **ql)
            except Exception as e:
                line["queue_c_abi_one_process"] = {"error": str(e)}
            del hostq
        emit(line)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
