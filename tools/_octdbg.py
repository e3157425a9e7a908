import sys, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from rumi_slam_amd.extractor import ORBextractor
from rumi_slam_amd.synth import synth_frame
from rumi_slam_amd import capi
ext = ORBextractor(1000, 1.2, 8, 20, 7, max_batch=1)
fr = torch.from_numpy(np.stack([synth_frame(9000)])).cuda()
L = capi.lib()
buf = (C.c_longlong * 128)()
for _ in range(3):
    ext.extract_batch(fr, (0, 1000), cap=1096); torch.cuda.synchronize()
    L.rumi_hook_oct_dbg(buf)
a = np.array(buf[:], dtype=np.int64)
a = a[a > 0]
print((np.diff(a)).tolist(), int(a[-1] - a[0]))
