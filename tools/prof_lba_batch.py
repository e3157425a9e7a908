"""rocprofv3 target: one warm local BA of BASELINE configs[3] and a batch of 16 such windows (rumi_local_ba_batch), after warm-up."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from ba_scene import ba_problem
from rumi_slam_amd.optimizer import Optimizer
opt = Optimizer()
b = ba_problem(seed=0, n_opt=20, n_fixed=5, n_points=3000)
a = (b["kf_pose"], b["kf_fixed"], b["mp_pos"], b["e_mp"], b["e_kf"], b["e_obs"], b["e_w"], b["K"])
R = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for _ in range(reps):
    t0 = time.perf_counter(); stats, *_ = opt.LocalBundleAdjustment(*a); print("lba ms", (time.perf_counter() - t0) * 1e3, "C call ms", opt.last_call_s * 1e3, stats, opt.stage_ms()[5])
if R > 0:
    for _ in range(reps):
        t0 = time.perf_counter(); r = opt.LocalBundleAdjustmentBatch([a] * R, 1); print("batch of %d: ms per window" % R, (time.perf_counter() - t0) * 1e3 / R, "C call ms per window", opt.last_call_s * 1e3 / R, r[0][0])
