import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from rumi_slam_amd.extractor import ORBextractor
from rumi_slam_amd.synth import synth_frame
B=1024
base = torch.from_numpy(np.stack([synth_frame(1234+i) for i in range(32)])).cuda()
fr = torch.stack([torch.roll(base[k % 32], shifts=(3*(k//32), 5*(k//32)), dims=(0,1)) for k in range(B)]).contiguous()
ext = ORBextractor(1000,1.2,8,20,7,max_batch=B)
host_pin = fr.cpu().pin_memory()
host_page = fr.cpu().numpy()
lst = [host_page[i] for i in range(B)]
for name, src in (("device", None), ("pinned", host_pin), ("pageable", lst)):
    for rep in range(3):
        torch.cuda.synchronize(); t=time.perf_counter()
        if src is None: ext.extract_batch(fr)
        else: ext.extract_batch_host(src)
        torch.cuda.synchronize(); dt=time.perf_counter()-t
    print(name, "%.2f ms  %.0f fps" % (dt*1e3, B/dt))
