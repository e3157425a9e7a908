#!/usr/bin/env python3
"""rumi_track_frame with and without the single-queue path must return the same step (different feature budgets, frames with few matches):
run once with RUMI_TRACK_SPECULATE=0 and once with 1, the two dumps must be equal.  usage: track_spec_check.py dump.npz"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import numpy as np
from rumi_slam_amd.extractor import ORBextractor
from rumi_slam_amd.tracker import Tracker
from rumi_slam_amd.synth import synth_frame
from test_track_frame_gpu import K_TUM3, PLANE_D, warp_homography, _homography, _pose_gt
out = {}
for nf in (500, 1000, 2000):
    ext = ORBextractor(nf, 1.2, 8, 20, 7)
    trk = Tracker(nf, 1.2, 8, 20, 7, 640, 480, 8192)
    sf = ext.GetScaleFactors()
    img0 = synth_frame(4242 + nf)
    fx, fy, cx, cy = K_TUM3.astype(np.float64)
    _, keys0, desc0 = ext(img0)
    n0 = len(keys0)
    pos = np.stack([(keys0["x"] - cx) / fx * PLANE_D, (keys0["y"] - cy) / fy * PLANE_D, np.full(n0, PLANE_D)], 1).astype(np.float32)
    dist0 = np.linalg.norm(pos, axis=1).astype(np.float32)
    lvl = keys0["octave"]
    pts = dict(pos=pos, normal=(pos / dist0[:, None]).astype(np.float32), max_dist=(dist0 * sf[lvl]).astype(np.float32),
               min_dist=(dist0 * sf[lvl] / sf[7]).astype(np.float32), desc=desc0.copy(), obs=np.ones(n0, np.int32), bad=np.zeros(n0, np.uint8),
               local=np.ones(n0, np.uint8))
    rng = np.random.default_rng(7)
    for frac in (0.5, 0.03, 0.0):                       # many matches; fewer than 20 (retry, giving up); none
        known = rng.random(n0) < frac
        last = dict(keys=keys0, mp=np.where(known, np.arange(n0), -1).astype(np.int32), outlier=np.zeros(n0, np.uint8))
        T = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
        for step in (1, 3):
            img = warp_homography(img0, _homography(*_pose_gt(step)))
            r = trk.track(img, K_TUM3, T, last["keys"], last["mp"], last["outlier"], pts, 15.0, 1.0)
            for k, v in r.items():
                out[f"{nf}_{frac}_{step}_{k}"] = np.asarray(v)
np.savez(sys.argv[1], **out)
print("cases", len(out))
