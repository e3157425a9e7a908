#!/usr/bin/env python3
"""Kernel-level view of the windowed matchers (run under rocprofv3 --kernel-trace --stats): 50 calls each of M1, M2, M3."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from rumi_slam_amd.matcher import FrameView, FeatureVector, ORBmatcher
from scene import K_TUM3, TrackingScene
s = TrackingScene(0)
m = ORBmatcher(0.8, True)
F = FrameView(s.cur_keys, s.cur_desc, s.w, s.h, s.sf)
mp = s.mappoint_view()
fm0 = np.full(F.n, -1, np.int32)
a2 = (s.Tcw7, K_TUM3, s.last_keys, s.last_mp, s.last_outlier, s.mp_pos, s.mp_desc, s.mp_obs, fm0)
fv1, fv2 = s.feature_vectors()
A, B = FeatureVector(fv1), FeatureVector(fv2)
KF = FrameView(s.last_keys, s.last_desc, s.w, s.h, s.sf)
bad = np.zeros(len(s.mp_obs), np.uint8)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
for name, fn in (("M1", lambda: m.SearchByProjection_MapPoints(F, mp, fm0, 3.0)), ("M2", lambda: m.SearchByProjection_Frame(F, *a2, 15.0)),
                 ("M3", lambda: m.SearchByBoW(KF, A, s.last_mp, bad, F, B))):
    if which not in ("all", name):
        continue
    fn()
    ts = []
    for _ in range(50):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    print(name, "us/call median", round(float(np.median(ts)) * 1e6, 1), "mean", round(float(np.mean(ts)) * 1e6, 1), "max", round(max(ts) * 1e6, 1))
