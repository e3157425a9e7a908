import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from ba_scene import ba_problem
from rumi_slam_amd.optimizer import Optimizer
opt = Optimizer()
b = ba_problem(seed=0, n_opt=20, n_fixed=5, n_points=3000)
a = (b["kf_pose"], b["kf_fixed"], b["mp_pos"], b["e_mp"], b["e_kf"], b["e_obs"], b["e_w"], b["K"])
for _ in range(6): opt.LocalBundleAdjustment(*a)
