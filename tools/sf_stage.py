import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from rumi_slam_amd.extractor import ORBextractor
from rumi_slam_amd.synth import synth_frame
ext = ORBextractor(1000, 1.2, 8, 20, 7, max_batch=1)
fr = torch.from_numpy(np.stack([synth_frame(9000)])).cuda()
for _ in range(3): ext.extract_batch(fr, (0, 1000), cap=1096)
ext.set_profiling(True)
for _ in range(3):
    ext.extract_batch(fr, (0, 1000), cap=1096); torch.cuda.synchronize()
    print({k: round(float(v) * 1e3, 1) for k, v in ext.stage_ms().items()})
