"""rumi_queue_extract, call by call: wall time of the Python call and the library's own split (extraction / exchange / total), for pageable and pinned record
buffers and pageable / pinned frames; 512 frames, one shard (RCCL on one rank) and two logical shards."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from rumi_slam_amd.queue import RuminationQueue
from rumi_slam_amd.synth import synth_frame
F = int(sys.argv[1]) if len(sys.argv) > 1 else 512
base = [synth_frame(1234 + i) for i in range(32)]
host = torch.from_numpy(np.stack([base[i % 32] if i < 32 else np.roll(base[i % 32], (7 * (i // 32), 11 * (i // 32)), (0, 1)) for i in range(F)]))
pin = host.pin_memory()
for shards in (1, 2):
    q = RuminationQueue(1000, 1.2, 8, 20, 7, [0] * shards, max_block=(F + shards - 1) // shards)
    for fname, fr in (("pinned frames", pin), ("pageable frames", host)):
        frames = [fr[f].numpy() for f in range(F)]
        for rname, rec in (("pageable records", np.zeros((F, q.record_bytes), np.uint8)), ("pinned records", torch.zeros((F, q.record_bytes), dtype=torch.uint8).pin_memory().numpy()),
                           ("no host records", None)):
            ts, lib = [], []
            for _ in range(8):
                t0 = time.perf_counter(); q.extract(frames, (0, 1000), want_host=rec is not None, out=rec); ts.append((time.perf_counter() - t0) * 1e3); lib.append(q.last_ms()["total"])
            print("%d shard(s), %s, %s: call ms %s | library ms %s -> %.1f k fps (median call)" % (shards, fname, rname, " ".join("%.2f" % t for t in ts[2:]), " ".join("%.2f" % t for t in lib[2:]),
                                                                                                    F / float(np.median(ts[2:]))))
    q.close()
