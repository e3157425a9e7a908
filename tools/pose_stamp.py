#!/usr/bin/env python3
"""One PoseOptimization call per size against the stamp build (tools/build_stamp_lib.sh): the kernel prints its own cycle accounting."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from ba_scene import pose_problem
from rumi_slam_amd.optimizer import Optimizer
opt = Optimizer()
for n in (64, 300, 600):
    p = pose_problem(5, n, 0.1)
    opt.PoseOptimization(p["Xw"], p["obs"], p["inv_sigma2"], p["K"], p["T0"])
