#!/usr/bin/env python3
"""Global bundle adjustment over a large window (block-sparse Schur + blocked Cholesky path): device time vs the CPU oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from ba_scene import ba_problem
from rumi_slam_amd.optimizer import Optimizer

opt = Optimizer(max_kf=512, max_mp=1 << 16, max_edges=1 << 20)
for cfg, its in ((dict(seed=32, n_opt=130, n_fixed=1, n_points=2500, outlier_frac=0.08), 6), (dict(seed=34, n_opt=170, n_fixed=1, n_points=6000), 10)):
    b = ba_problem(**cfg)
    a = (b["kf_pose"], b["kf_fixed"], b["mp_pos"], b["e_mp"], b["e_kf"], b["e_obs"], b["e_w"], b["K"])
    t0 = time.perf_counter(); its_ref, kp_ref, mp_ref = O.bundle_adjustment(*a, its, True); tc = time.perf_counter() - t0
    opt.BundleAdjustment(*a, n_iterations=its, robust=True)
    t0 = time.perf_counter(); stats, kp, mp = opt.BundleAdjustment(*a, n_iterations=its, robust=True); tg = time.perf_counter() - t0
    print(cfg, "edges", len(b["e_mp"]), "iterations ref/gpu", its_ref, stats[0], "trials", stats[1], "cpu oracle %.1f ms, gpu wall %.2f ms, device %.2f ms" % (tc * 1e3, tg * 1e3, opt.stage_ms()[5]),
          "max pose diff %.2e" % np.abs(kp - kp_ref).max())
