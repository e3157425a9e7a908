#!/usr/bin/env python3
"""Latency of SearchByBoW against K relocalisation candidates: one rumi_search_by_bow_batch call, K single calls, K oracle calls (host arrays in / out)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from scene import TrackingScene
from rumi_slam_amd.matcher import ORBmatcher, FrameView, FeatureVector, SearchByBoW_batch


def fv(d):
    return FeatureVector(d)


for K, nodes in ((10, 600), (10, 100), (1, 600)):
    scenes = [TrackingScene(40 + k) for k in range(K)]
    base = scenes[0]
    F = FrameView(base.cur_keys, base.cur_desc, base.w, base.h, base.sf)
    fkf0, ff0 = base.feature_vectors(nodes)
    b = fv(ff0)
    KFs, fvs, mps, bads = [], [], [], []
    for k, s in enumerate(scenes):
        a, _ = s.feature_vectors(nodes)
        KFs.append(FrameView(s.last_keys, s.last_desc, s.w, s.h, s.sf)); fvs.append(fv(a)); mps.append(s.last_mp)
        bads.append(np.zeros(len(s.mp_obs), np.uint8))
    m = ORBmatcher(0.75, True)
    def med(f, n=30):
        f(); t = []
        for _ in range(n):
            t0 = time.perf_counter(); f(); t.append(time.perf_counter() - t0)
        return float(np.median(t)) * 1e6
    tb = med(lambda: SearchByBoW_batch(m, KFs, fvs, mps, bads, F, b))
    ts = med(lambda: [m.SearchByBoW(KFs[k], fvs[k], mps[k], bads[k], F, b) for k in range(K)])
    to = med(lambda: [O.search_by_bow(KFs[k].keys, KFs[k].desc, mps[k], bads[k], (fvs[k].node_ids, fvs[k].offsets, fvs[k].indices), base.cur_keys,
                                      base.cur_desc, (b.node_ids, b.offsets, b.indices), 0.75, True) for k in range(K)], 10)
    print(f"K={K} nodes={nodes}: batch {tb:.0f} us, {K} single calls {ts:.0f} us, {K} oracle calls {to:.0f} us")
