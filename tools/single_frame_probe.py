#!/usr/bin/env python3
"""Latency of the single-frame host API (rumi_orb_extract): image in host memory -> key-points + descriptors in host memory."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from rumi_slam_amd.extractor import ORBextractor
from rumi_slam_amd.synth import synth_frame
from tools.bench_matrix import median_call
ext = ORBextractor(1000, 1.2, 8, 20, 7)
imgs = [synth_frame(9000 + i) for i in range(4)]
r0 = [ext(im) for im in imgs]
r1 = [ext(im) for im in imgs]
assert all(a[0] == b[0] and a[1].tobytes() == b[1].tobytes() and np.array_equal(a[2], b[2]) for a, b in zip(r0, r1))
k = [0]
def one():
    k[0] += 1
    return ext(imgs[k[0] & 3])
print("single frame median %.1f us" % (median_call(one, 200) * 1e6))
pin = ext.image_buffer(imgs[0].shape[1], imgs[0].shape[0])          # the frame captured straight into the handle's pinned staging memory
pin[:] = imgs[0]
rp = ext(pin)
assert rp[0] == r0[0][0] and rp[1].tobytes() == r0[0][1].tobytes() and np.array_equal(rp[2], r0[0][2])
print("single frame captured into rumi_orb_image_buffer's memory, median %.1f us" % (median_call(lambda: ext(pin), 200) * 1e6))
import time
real, spent = ext._lib.rumi_orb_extract, []
def timed(*a):
    t0 = time.perf_counter(); rc = real(*a); spent.append(time.perf_counter() - t0); return rc
ext._lib.rumi_orb_extract = timed                                   # the C entry alone, without the Python mirror's array allocations and copies
for _ in range(205): ext(pin)
ext._lib.rumi_orb_extract = real
print("  the C call alone (pinned capture buffer): median %.1f us" % (float(np.median(spent[5:])) * 1e6))
# the same frame already in device memory, results left in device memory (no transfers: the kernels' dependent chain alone)
import torch
fr = torch.from_numpy(imgs[0]).cuda()[None].contiguous()
ext1 = ORBextractor(1000, 1.2, 8, 20, 7, max_batch=1)
obuf = (torch.empty((1, 1096, 7), dtype=torch.float32, device="cuda"), torch.empty((1, 1096, 32), dtype=torch.uint8, device="cuda"),
        torch.zeros((1, 2), dtype=torch.int32, device="cuda"))
def dev():
    ext1.extract_batch(fr, (0, 1000), cap=1096, out=obuf)      # (waits for the results: the blocking form of the call)
print("single frame, device-resident in and out, median %.1f us" % (median_call(dev, 200) * 1e6))
