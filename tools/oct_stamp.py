#!/usr/bin/env python3
"""Scratch probe: one-frame extraction with the octree kernel's cycle stamps (library built with -DRUMI_OCT_STAMP prints them)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rumi_slam_amd.capi as capi
if os.environ.get('RUMI_STAMP_LIB'):
    capi.LIB_PATH = os.environ['RUMI_STAMP_LIB']
from rumi_slam_amd.extractor import ORBextractor
from rumi_slam_amd.synth import synth_frame
fr = torch.from_numpy(synth_frame(1234)).cuda()[None].contiguous()
ext = ORBextractor(1000, 1.2, 8, 20, 7, max_batch=1)
for i in range(3):
    ext.extract_batch(fr, (0, 1000), cap=1096)
    torch.cuda.synchronize()
    print("---", flush=True)
ext.close()
