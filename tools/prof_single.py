#!/usr/bin/env python3
"""Single-frame host API in a loop (for rocprofv3 --kernel-trace: launch gaps of the real-time path)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rumi_slam_amd.extractor import ORBextractor
from rumi_slam_amd.synth import synth_frame
ext = ORBextractor(1000, 1.2, 8, 20, 7)
img = synth_frame(1234)
ext(img)
ts = []
for _ in range(60):
    t0 = time.perf_counter(); ext(img); ts.append(time.perf_counter() - t0)
print("median us", np.median(ts) * 1e6)
