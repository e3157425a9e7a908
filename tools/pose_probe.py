import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import oracle_lib as O
from ba_scene import pose_problem
from rumi_slam_amd.optimizer import Optimizer
from tools.bench_matrix import median_call
opt = Optimizer()
for n in (150, 300, 600):
    p = pose_problem(5, n, 0.1)
    a = (p["Xw"], p["obs"], p["inv_sigma2"], p["K"], p["T0"])
    r = O.pose_optimization(*a); g = opt.PoseOptimization(*a)
    assert r[0] == g[0] and np.array_equal(r[2], g[2])
    print(n, "edges", len(p["obs"]), "gpu us %.1f cpu us %.1f" % (median_call(lambda: opt.PoseOptimization(*a), 50) * 1e6, median_call(lambda: O.pose_optimization(*a), 10) * 1e6))
