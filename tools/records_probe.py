#!/usr/bin/env python3
"""128-frame steps: arrays vs records output, matching on its own stream vs on the caller's, with / without the (degenerate) exchange."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rumi_slam_amd.extractor import ORBextractor
from rumi_slam_amd.matcher import bruteforce_ring
from rumi_slam_amd import rumination
from rumi_slam_amd.synth import synth_frame
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 128
host = np.stack([synth_frame(1234 + i) for i in range(32)])
base = torch.from_numpy(host).cuda()
fr = torch.empty((nb, 480, 640), dtype=torch.uint8, device="cuda")
for k in range(nb):
    fr[k] = torch.roll(base[k % 32], shifts=(7 * (k // 32), 11 * (k // 32)), dims=(0, 1)) if k >= 32 else base[k]
ext = ORBextractor(1000, 1.2, 8, 20, 7, max_batch=nb)
ext.set_resident_queue(True)
cap = 1096
ms = torch.cuda.Stream()
NB = 4
arr = [(torch.empty((nb, cap, 7), dtype=torch.float32, device='cuda'), torch.empty((nb, cap, 32), dtype=torch.uint8, device='cuda'), torch.zeros((nb, 2), dtype=torch.int32, device='cuda')) for _ in range(NB)]
rec = [torch.zeros((nb, rumination.record_bytes(cap)), dtype=torch.uint8, device='cuda') for _ in range(NB)]
views = [rumination.record_views(r, cap) for r in rec]
torch.cuda.synchronize()
def run(records, mstream, match=True):
    cons = [None] * NB
    i = [0]
    def step():
        k = i[0] % NB; i[0] += 1
        if cons[k] is not None: ext.wait_event(cons[k])
        if records:
            ext.extract_batch_records(fr, (0, 1000), cap=cap, wait=False, out=rec[k]); kp, d, c = views[k]
        else:
            kp, d, c = ext.extract_batch(fr, (0, 1000), cap=cap, wait=False, out=arr[k])
        if not match: return
        if mstream:
            ev = torch.cuda.Event(); ev.record()
            with torch.cuda.stream(ms):
                ms.wait_event(ev)
                bruteforce_ring(d, c)
                cons[k] = torch.cuda.Event(); cons[k].record(ms)
        else:
            bruteforce_ring(d, c)
    for _ in range(6): step()
    ext.sync(); torch.cuda.synchronize()
    best = 0
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(40): step()
        ext.sync(); torch.cuda.synchronize()
        best = max(best, nb * 40 / (time.perf_counter() - t0))
    return best / 1e3
for records in (0, 1):
    print(nb, "records" if records else "arrays ", "extract only %.1fk   match same stream %.1fk   match own stream %.1fk" % (run(records, 0, False), run(records, 0), run(records, 1)))
