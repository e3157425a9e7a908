#!/usr/bin/env python3
"""Debug aid: FAST candidate lists of the GPU path against the oracle, level by level, with the first differences."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
from rumi_slam_amd.extractor import ORBextractor
from rumi_slam_amd.synth import synth_frame
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1234
img = synth_frame(seed)
g = ORBextractor(1000, 1.2, 8, 20, 7)
o = oracle_lib.OracleExtractor(1000, 1.2, 8, 20, 7)
g(img, None, (0, 1000)); o.extract(img, (0, 1000))
for l in range(8):
    gc, oc = g.stage_keypoints(l, 0), o.keypoints(l, False)
    gs = set((float(k['x']), float(k['y']), float(k['response'])) for k in gc)
    os_ = set((float(k['x']), float(k['y']), float(k['response'])) for k in oc)
    print(f"level {l}: gpu {len(gc)} oracle {len(oc)} only-gpu {len(gs - os_)} only-oracle {len(os_ - gs)} same-order {gc.tobytes() == oc.tobytes()}")
    for t in sorted(gs - os_)[:8]: print("   only gpu   ", t)
    for t in sorted(os_ - gs)[:8]: print("   only oracle", t)
