"""Randomised parity soak of the extractor against the CPU oracle (run on the GPU box): many frames, varied texture / noise / size /
feature count, through the single-frame call (quadtree with keys in registers), a small batch (<= 32 frames: same kernel variant) and
a large batch (keys-in-memory variant, pipelined sub-chunks).  Prints one line per configuration and the number of mismatching frames.
usage: python tools/soak_extractor.py [frames_per_config]"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib as O
from rumi_slam_amd.extractor import ORBextractor
from rumi_slam_amd.synth import synth_frame

N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
CONFIGS = [(640, 480, 1000), (640, 480, 2000), (752, 480, 1000), (600, 350, 1000), (640, 480, 500), (1241, 376, 2000), (320, 240, 1000), (640, 480, 5000)]
bad_total = 0
for ci, (w, h, nf) in enumerate(CONFIGS):
    rng = np.random.default_rng(1000 + ci)
    frames = []
    for i in range(N):
        kind = i % 4
        if kind == 0: img = synth_frame(50000 + ci * 1000 + i, w=w, h=h)
        elif kind == 1: img = synth_frame(50000 + ci * 1000 + i, w=w, h=h, n_rect=int(rng.integers(5, 150)), noise=int(rng.integers(0, 3)), contrast=(8, 40))
        elif kind == 2: img = synth_frame(50000 + ci * 1000 + i, w=w, h=h, n_rect=int(rng.integers(300, 900)), noise=int(rng.integers(2, 12)))
        else:
            img = synth_frame(50000 + ci * 1000 + i, w=w, h=h, n_rect=200).astype(np.int32)
            img[:, : w // 2] = img[:, : w // 2] // 3 + 5           # a dark half: candidates only from the minThFAST retry there
            img = np.clip(img, 0, 255).astype(np.uint8)
        frames.append(img)
    host = np.stack(frames)
    t0 = time.time()
    orcs = [O.OracleExtractor(nf, 1.2, 8, 20, 7) for _ in range(8)]
    def ref(k):
        return [orcs[k].extract(host[i], (0, 1000)) for i in range(k, N, 8)]
    with ThreadPoolExecutor(8) as ex:
        parts = list(ex.map(ref, range(8)))
    want = [None] * N
    for k in range(8):
        for j, i in enumerate(range(k, N, 8)): want[i] = parts[k][j]
    tcpu = time.time() - t0
    cap = nf + 4 * 8 + 64
    dev = torch.from_numpy(host).cuda()
    def same(got, i):
        mono, kp, desc = want[i]
        gm, gk, gd = got
        return gm == mono and len(gk) == len(kp) and gk.tobytes() == kp.tobytes() and np.array_equal(gd, desc)
    bad = {"single": 0, "batch8": 0, "batchN": 0}
    e1 = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
    for i in range(N):
        bad["single"] += not same(e1(host[i], None, (0, 1000)), i)
    for name, B in (("batch8", 8), ("batchN", N)):
        eb = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B)
        for b0 in range(0, N, B):
            kp, desc, counts = eb.extract_batch(dev[b0:b0 + B], (0, 1000), cap=cap)
            torch.cuda.synchronize()
            kp, desc, counts = kp.cpu().numpy(), desc.cpu().numpy(), counts.cpu().numpy()
            for j in range(min(B, N - b0)):
                n, mono = int(counts[j, 0]), int(counts[j, 1])
                rec = np.ascontiguousarray(kp[j, :n]).view(want[b0 + j][1].dtype).reshape(-1)
                bad[name] += not same((mono, rec, desc[j, :n]), b0 + j)
    bad_total += sum(bad.values())
    print(f"{w}x{h} nf={nf}: {N} frames, mean kp {np.mean([len(x[1]) for x in want]):.0f}, oracle {tcpu:.1f}s, mismatching frames {bad}", flush=True)
print("TOTAL MISMATCHES", bad_total)
sys.exit(1 if bad_total else 0)
