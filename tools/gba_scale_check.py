#!/usr/bin/env python3
"""One-off scale check of the large-window bundle adjustment: 400 optimised key-frames (2400 unknowns in the reduced system) against the
CPU oracle's dense solve.  Slow on the CPU side (minutes); not part of the test suite."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from ba_scene import ba_problem
from rumi_slam_amd.optimizer import Optimizer

K = int(sys.argv[1]) if len(sys.argv) > 1 else 400
t0 = time.perf_counter(); b = ba_problem(seed=77, n_opt=K, n_fixed=1, n_points=20 * K); print("scene %.1f s, edges %d" % (time.perf_counter() - t0, len(b["e_mp"])), flush=True)
a = (b["kf_pose"], b["kf_fixed"], b["mp_pos"], b["e_mp"], b["e_kf"], b["e_obs"], b["e_w"], b["K"])
opt = Optimizer(max_kf=K + 8, max_mp=20 * K + 8, max_edges=len(b["e_mp"]) + 8)
opt.BundleAdjustment(*a, n_iterations=3, robust=True)
t0 = time.perf_counter(); stats, kp, mp = opt.BundleAdjustment(*a, n_iterations=3, robust=True); tg = time.perf_counter() - t0
print("gpu: iterations %d trials %d wall %.1f ms device %.1f ms" % (stats[0], stats[1], tg * 1e3, opt.stage_ms()[5]), flush=True)
t0 = time.perf_counter(); its_ref, kp_ref, mp_ref = O.bundle_adjustment(*a, 3, True); tc = time.perf_counter() - t0
rel = np.linalg.norm(mp - mp_ref, axis=1) / np.maximum(np.linalg.norm(mp_ref, axis=1), 1e-3)
print("cpu oracle: iterations %d, %.1f s; max pose diff %.2e, max landmark rel diff %.2e" % (its_ref, tc, np.abs(kp - kp_ref).max(), rel.max()))
