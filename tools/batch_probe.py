#!/usr/bin/env python3
"""Frames per second of device-resident extract (+ match) at small batch sizes: back-to-back asynchronous calls, one final sync.
usage: batch_probe.py [nb ...]   (env: RUMI_PARTS, RUMI_SUBMAX)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rumi_slam_amd.extractor import ORBextractor
from rumi_slam_amd.matcher import bruteforce_ring
from rumi_slam_amd.synth import synth_frame
sizes = [int(a) for a in sys.argv[1:]] or [64, 128, 256, 1024]
B = max(sizes)
host = np.stack([synth_frame(1234 + i) for i in range(32)])
base = torch.from_numpy(host).cuda()
fr = torch.empty((B, 480, 640), dtype=torch.uint8, device="cuda")
for k in range(B):
    fr[k] = torch.roll(base[k % 32], shifts=(7 * (k // 32), 11 * (k // 32)), dims=(0, 1)) if k >= 32 else base[k]
ext = ORBextractor(1000, 1.2, 8, 20, 7, max_batch=B)
if os.environ.get('RUMI_RESIDENT'): ext.set_resident_queue(True)
mstream = torch.cuda.Stream()
cap = 1096
res = {}
for nb in sizes:
    sub = fr[:nb]
    obuf = [(torch.empty((nb, cap, 7), dtype=torch.float32, device='cuda'), torch.empty((nb, cap, 32), dtype=torch.uint8, device='cuda'), torch.empty((nb, 2), dtype=torch.int32, device='cuda')) for _ in range(2)]
    torch.cuda.synchronize(); cnt = [0]
    for withm in (0, 1, 2):
        def step():
            cnt[0] += 1
            kp, desc, counts = ext.extract_batch(sub, (0, 1000), cap=cap, wait=False, out=obuf[cnt[0] & 1])
            if withm == 1:
                return bruteforce_ring(desc, counts)
            if withm == 2:                                    # the matching on a stream of its own behind the extraction, as bench.py's step does
                ev = torch.cuda.Event(); ev.record()
                with torch.cuda.stream(mstream):
                    mstream.wait_event(ev)
                    m = bruteforce_ring(desc, counts)
                desc.record_stream(mstream); counts.record_stream(mstream)
                return m
        reps = max(6, 4096 // nb)
        for _ in range(3): step()
        ext.sync(); torch.cuda.synchronize()
        best = 0
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(reps): step()
            ext.sync(); torch.cuda.synchronize()
            best = max(best, nb * reps / (time.perf_counter() - t0))
        res[(nb, withm)] = best
print("PARTS", os.environ.get("RUMI_PARTS"), "SUBMAX", os.environ.get("RUMI_SUBMAX"), "RESIDENT", os.environ.get("RUMI_RESIDENT"), " ".join(f"{nb}{'ems'[m]}={v/1e3:.1f}k" for (nb, m), v in res.items()))
