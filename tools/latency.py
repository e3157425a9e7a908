"""Single-frame (host buffers in / out, PCIe included) latency of the drop-in calls; prints JSON."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from rumi_slam_amd.extractor import ORBextractor
from rumi_slam_amd.synth import synth_frame
ext = ORBextractor(1000, 1.2, 8, 20, 7)
imgs = [synth_frame(10 + i) for i in range(8)]
for im in imgs: ext(im)
t0 = time.perf_counter(); n = 0
for r in range(10):
    for im in imgs: ext(im); n += 1
dt = (time.perf_counter() - t0) / n
print(json.dumps({"orb_extract_single_frame_ms_incl_pcie": round(dt * 1e3, 4), "fps": round(1 / dt, 1)}))
