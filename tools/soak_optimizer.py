"""Randomised parity soak of the optimisers against the CPU oracle (run on the GPU box): many PoseOptimization problems (single and
batched), local BA windows of varied shape, OptimizeSim3 / OptimizeCloudSim3 problems.  Reports the worst relative deviation per family
and counts every case outside the 1e-4 bar or with a differing integer result (outlier / erase flags, iteration counts, inlier counts).
usage: python tools/soak_optimizer.py [scale]   (scale 1: ~400 pose problems, 24 windows, 60 Sim3 problems)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from ba_scene import ba_problem, pose_problem
from sim3_scene import sim3_pair_problem, sim3_cloud_problem
from rumi_slam_amd.optimizer import Optimizer

S = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
opt = Optimizer()
RTOL = 1e-4
bad = 0


def pose_dev(a, b):
    return max(np.linalg.norm(a[:4] - b[:4]) / max(1.0, np.linalg.norm(b[:4])), np.linalg.norm(a[4:] - b[4:]) / max(1e-2, np.linalg.norm(b[4:])))


rng = np.random.default_rng(5)
worst, nbad, t0 = 0.0, 0, time.time()
probs = []
for i in range(int(400 * S)):
    p = pose_problem(1000 + i, int(rng.integers(8, 1500)), float(rng.choice([0.0, 0.05, 0.1, 0.2, 0.35])))
    ref = O.pose_optimization(p["Xw"], p["obs"], p["inv_sigma2"], p["K"], p["T0"])
    got = opt.PoseOptimization(p["Xw"], p["obs"], p["inv_sigma2"], p["K"], p["T0"])
    d = pose_dev(got[1], ref[1]); worst = max(worst, d)
    nbad += (got[0] != ref[0]) or not np.array_equal(got[2], ref[2]) or d > RTOL
    probs.append((p, ref))
print(f"PoseOptimization: {len(probs)} problems, worst relative pose deviation {worst:.2e}, outside the bar {nbad}  ({time.time() - t0:.0f}s)", flush=True)
bad += nbad
nbad, worst = 0, 0.0
for g0 in range(0, len(probs), 64):
    grp = probs[g0:g0 + 64]
    start = np.cumsum([0] + [len(p["inv_sigma2"]) for p, _ in grp]).astype(np.int32)
    ng, T, out = opt.PoseOptimizationBatch(start, np.concatenate([p["Xw"] for p, _ in grp]), np.concatenate([p["obs"] for p, _ in grp]),
                                           np.concatenate([p["inv_sigma2"] for p, _ in grp]), grp[0][0]["K"], np.stack([p["T0"] for p, _ in grp]))
    for i, (p, ref) in enumerate(grp):
        d = pose_dev(T[i], ref[1]); worst = max(worst, d)
        nbad += (ng[i] != ref[0]) or not np.array_equal(out[start[i]:start[i + 1]], ref[2]) or d > RTOL
print(f"PoseOptimizationBatch: worst {worst:.2e}, outside the bar {nbad}", flush=True)
bad += nbad

nbad, worst, t0 = 0, 0.0, time.time()
n_win = int(24 * S)
for i in range(n_win):
    cfg = dict(seed=300 + i, n_opt=int(rng.integers(1, 36)), n_fixed=int(rng.integers(1, 8)), n_points=int(rng.integers(100, 3500)),
               outlier_frac=float(rng.choice([0.0, 0.05, 0.15])))
    b = ba_problem(**cfg)
    a = (b["kf_pose"], b["kf_fixed"], b["mp_pos"], b["e_mp"], b["e_kf"], b["e_obs"], b["e_w"], b["K"])
    its_ref, kp_ref, mp_ref, er_ref = O.local_ba(*a)
    stats, kp, mp, er = opt.LocalBundleAdjustment(*a)
    d = max([pose_dev(kp[k], kp_ref[k]) for k in range(len(kp))] + [float((np.linalg.norm(mp - mp_ref, axis=1) / np.maximum(np.linalg.norm(mp_ref, axis=1), 1e-2)).max())])
    worst = max(worst, d)
    wrong = stats[0] != its_ref or np.count_nonzero(er != er_ref) or d > RTOL
    if wrong: print("  LBA outside the bar:", cfg, "iterations", stats[0], its_ref, "erase diffs", int(np.count_nonzero(er != er_ref)), f"dev {d:.2e}")
    nbad += bool(wrong)
print(f"LocalBundleAdjustment: {n_win} windows, worst relative deviation {worst:.2e}, outside the bar {nbad}  ({time.time() - t0:.0f}s)", flush=True)
bad += nbad

nbad, worst, t0 = 0, 0.0, time.time()
n_s = int(30 * S)
for i in range(n_s):
    b = sim3_pair_problem(seed=500 + i, n=int(rng.integers(30, 800)), scale=float(rng.uniform(0.7, 1.5)), outlier_frac=float(rng.choice([0.0, 0.1, 0.3])))
    a = (b["S0"], b["P1c"], b["P2c"], b["obs1"], b["obs2"], b["w1"], b["w2"], b["K"], b["K"], 10.0, bool(i % 3 == 0), True)
    ref, got = O.optimize_sim3(*a), opt.OptimizeSim3(*a)
    d = float(np.abs(np.asarray(got[3]) - np.asarray(ref[3])).max() / max(1.0, np.abs(np.asarray(ref[3])).max())); worst = max(worst, d)
    nbad += got[:3] != ref[:3] or not np.array_equal(got[4], ref[4]) or d > RTOL
    c = sim3_cloud_problem(seed=700 + i, n_pairs=int(rng.integers(2, 12)), per_pair=int(rng.integers(40, 250)))
    a = (c["S0"], c["P1c"], c["P2c"], c["obs1"], c["obs2"], c["w1"], c["w2"], c["K"], c["K"], 10.0, True, False, c["pair_of"], c["S_c1w"], c["S_c2w"], c["skip12"], c["skip21"])
    ref, got = O.optimize_sim3(*a), opt.OptimizeSim3(*a)
    d = float(np.abs(np.asarray(got[3]) - np.asarray(ref[3])).max() / max(1.0, np.abs(np.asarray(ref[3])).max())); worst = max(worst, d)
    nbad += got[:3] != ref[:3] or not np.array_equal(got[4], ref[4]) or d > RTOL
print(f"OptimizeSim3 / OptimizeCloudSim3: {2 * n_s} problems, worst relative deviation {worst:.2e}, outside the bar {nbad}  ({time.time() - t0:.0f}s)", flush=True)
bad += nbad
print("TOTAL OUTSIDE THE BAR", bad)
sys.exit(1 if bad else 0)
