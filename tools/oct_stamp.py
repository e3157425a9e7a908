#!/usr/bin/env python3
"""One-frame extraction with the quadtree kernel's in-kernel cycle stamps (a library built by tools/build_stamp_lib.sh prints them:
   python tools/with_lib.py tools/bin/librumi_hip_stamp.so tools/oct_stamp.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rumi_slam_amd.extractor import ORBextractor
from rumi_slam_amd.synth import synth_frame
fr = torch.from_numpy(synth_frame(1234)).cuda()[None].contiguous()
ext = ORBextractor(1000, 1.2, 8, 20, 7, max_batch=1)
for i in range(3):
    ext.extract_batch(fr, (0, 1000), cap=1096)
    torch.cuda.synchronize()
    print("---", flush=True)
ext.close()
