#!/usr/bin/env python3
"""One-frame extraction with the quadtree kernel's in-kernel cycle stamps (a library built by tools/build_stamp_lib.sh prints them:
   python tools/with_lib.py tools/bin/librumi_hip_stamp.so tools/oct_stamp.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rumi_slam_amd.extractor import ORBextractor
from rumi_slam_amd.synth import synth_frame
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "diagonal":       # corners in a thin diagonal band: the tree outgrows its count tables
    img = np.full((480, 640), 120, np.uint8)
    yy, xx = np.mgrid[0:480, 0:640]
    band = np.abs(yy - 0.73 * xx - 5) < 9
    img[band] = np.random.default_rng(40).integers(0, 256, int(band.sum()), dtype=np.uint8)
else:
    img = synth_frame(1234)
fr = torch.from_numpy(img).cuda()[None].contiguous()
ext = ORBextractor(1000, 1.2, 8, 20, 7, max_batch=1)
for i in range(3):
    ext.extract_batch(fr, (0, 1000), cap=1096)
    torch.cuda.synchronize()
    print("---", flush=True)
ext.close()
