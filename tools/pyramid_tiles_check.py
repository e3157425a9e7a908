#!/usr/bin/env python3
"""The one-launch pyramid of few-frame calls (k_pyramid_tiles) against the launch-per-level chain: run once with RUMI_PYRAMID_TILES=0 and once
with =1, the two dumps (every level of every frame, several image sizes, scale factors and level counts, calls of 1 and 3 frames) must be
equal byte for byte.  usage: pyramid_tiles_check.py dump.npz"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
from rumi_slam_amd.extractor import ORBextractor
out = {}
rng = np.random.default_rng(11)
for (w, h, sf, nl) in ((640, 480, 1.2, 8), (752, 480, 1.2, 8), (600, 350, 1.2, 8), (320, 240, 1.2, 8), (641, 479, 1.2, 8), (640, 480, 1.1, 8),
                       (640, 480, 1.5, 5), (1280, 720, 1.2, 8), (640, 480, 1.2, 2)):
    for nb in (1, 3):
        coarse = rng.integers(0, 256, (nb, h // 16 + 1, w // 16 + 1)).astype(np.int32)      # blocks of 16 x 16 plus a little noise: every byte matters to the resize, few corners
        frames = np.clip(np.kron(coarse, np.ones((16, 16), np.int32))[:, :h, :w] + rng.integers(-3, 4, (nb, h, w)), 0, 255).astype(np.uint8)
        ext = ORBextractor(500, sf, nl, 20, 7, max_width=w, max_height=h, max_batch=nb)
        ext.extract_batch(torch.from_numpy(frames).cuda(), (0, 1000))
        for f in range(nb):
            for l in range(1, nl):
                out[f"{w}x{h}_sf{sf}_nl{nl}_b{nb}_f{f}_l{l}"] = ext.pyramid_level(l, frame=f)
np.savez(sys.argv[1], **out)
print(len(out), "levels dumped")
