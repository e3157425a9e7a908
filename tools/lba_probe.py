"""Local BA of BASELINE configs[3] (20 + 5 key-frames x 3000 points): wall / device ms and per-kernel event times per trial, for both solve kernels
(RUMI_BA_SOLVE_PANEL8=1 selects the older one in a fresh process)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from ba_scene import ba_problem
from rumi_slam_amd.optimizer import Optimizer
opt = Optimizer()
for n_opt in (int(x) for x in (sys.argv[1:] or ["20"])):
    b = ba_problem(seed=0, n_opt=n_opt, n_fixed=5, n_points=3000)
    a = (b["kf_pose"], b["kf_fixed"], b["mp_pos"], b["e_mp"], b["e_kf"], b["e_obs"], b["e_w"], b["K"])
    for _ in range(3): opt.LocalBundleAdjustment(*a)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); stats, *_ = opt.LocalBundleAdjustment(*a); ts.append((time.perf_counter() - t0) * 1e3)
    dev = opt.stage_ms()[5]
    print("n_opt %d  wall ms median %.3f min %.3f  device %.3f  stats %s" % (n_opt, float(np.median(ts)), min(ts), dev, [int(x) for x in stats]))

# R windows through rumi_local_ba_batch
b = ba_problem(seed=0, n_opt=20, n_fixed=5, n_points=3000)
w = (b["kf_pose"], b["kf_fixed"], b["mp_pos"], b["e_mp"], b["e_kf"], b["e_obs"], b["e_w"], b["K"])
for R, workers in ((16, 1), (32, 1), (4, 1)):
    opt.LocalBundleAdjustmentBatch([w] * R, workers)
    dts, cpu = [], []
    for _ in range(5):
        c0, t0 = time.process_time(), time.perf_counter(); opt.LocalBundleAdjustmentBatch([w] * R, workers); dts.append(time.perf_counter() - t0); cpu.append(time.process_time() - c0)
    print("batch of %d windows (20 + 5 key-frames x 3000 points), %d workers: %.3f ms per window (best of 5; median %.3f); host cores busy %.2f (process CPU time / wall time, LM loop %s)" % (
        R, workers, min(dts) * 1e3 / R, float(np.median(dts)) * 1e3 / R, float(np.median(cpu)) / float(np.median(dts)), "on the HOST (RUMI_BA_HOST_LM)" if os.environ.get("RUMI_BA_HOST_LM") else "on the device"))
