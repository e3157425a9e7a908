#!/usr/bin/env python3
"""Per-stage device time of the batched extractor for several feature counts (HIP events on the kernels' streams)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rumi_slam_amd.extractor import ORBextractor
from rumi_slam_amd.synth import synth_frame
B = 256
uniq = np.stack([synth_frame(1234 + i) for i in range(32)])
fr = torch.from_numpy(uniq).cuda().repeat(B // 32, 1, 1).contiguous()
for nf in [int(a) for a in sys.argv[1:]] or [1000, 2000, 5000]:
    ext = ORBextractor(nf, 1.2, 8, 20, 7, max_batch=B)
    for _ in range(2):
        ext.extract_batch(fr, (0, 1000), cap=nf + 96)
    ext.set_profiling(True)
    ext.extract_batch(fr, (0, 1000), cap=nf + 96)
    torch.cuda.synchronize()
    print(nf, {k: round(float(v), 3) for k, v in ext.stage_ms().items()})
    ext.close()
