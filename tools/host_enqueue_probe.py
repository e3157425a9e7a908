#!/usr/bin/env python3
"""Host-side cost of enqueueing one bench.py step (no device wait inside the loop), per component, for small batches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rumi_slam_amd.extractor import ORBextractor
from rumi_slam_amd.matcher import bruteforce_ring
from rumi_slam_amd.synth import synth_frame
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
host = np.stack([synth_frame(1234 + i) for i in range(32)])
fr = torch.from_numpy(host).cuda().repeat((nb + 31) // 32, 1, 1)[:nb].contiguous()
ext = ORBextractor(1000, 1.2, 8, 20, 7, max_batch=nb)
ext.set_resident_queue(True)
cap = 1096
out = [(torch.empty((nb, cap, 7), dtype=torch.float32, device='cuda'), torch.empty((nb, cap, 32), dtype=torch.uint8, device='cuda'), torch.empty((nb, 2), dtype=torch.int32, device='cuda')) for _ in range(4)]
torch.cuda.synchronize()
def t(fn, n=200):
    for _ in range(20): fn()
    ext.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = (time.perf_counter() - t0) / n
    ext.sync(); torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / n
    return dt * 1e6, tot * 1e6
i = [0]
def ex():
    i[0] += 1
    return ext.extract_batch(fr, (0, 1000), cap=cap, wait=False, out=out[i[0] & 3])
def exm():
    kp, d, c = ex()
    bruteforce_ring(d, c)
print("extract only: host %.0f us / step, total %.0f us / step" % t(ex))
print("extract+match same stream: host %.0f us, total %.0f us" % t(exm))
