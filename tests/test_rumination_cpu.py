"""world_size-2 gloo test of the rumination sharding + all-gather (the N > 1 path of bench.py); extraction itself is stood in
for by the CPU oracle here (no GPU in this container) — what is under test is the partition and the exchange step."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from rumi_slam_amd import rumination
from rumi_slam_amd.synth import synth_frame

CAP = 600


def _records(frames):
    ext = O.OracleExtractor(500, 1.2, 8, 20, 7)
    b = len(frames)
    kp = np.zeros((b, CAP, 7), np.float32); desc = np.zeros((b, CAP, 32), np.uint8); counts = np.zeros((b, 2), np.int32)
    for i, f in enumerate(frames):
        mono, k, d = ext.extract(f, (0, 1000))
        counts[i] = (len(k), mono)
        kp[i, :len(k)] = k.view(np.float32).reshape(-1, 7)
        desc[i, :len(k)] = d
    return torch.from_numpy(kp), torch.from_numpy(desc), torch.from_numpy(counts)


def _worker(rank, world, port, n_frames, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = rumination.shard_bounds(n_frames, rank, world)
    frames = [synth_frame(300 + i, w=320, h=240) for i in range(lo, hi)]
    counts, kp, desc = rumination.extract_queue(_records, frames, n_frames)
    # the pipelined form bench.py uses: two exchanges in flight order, joined later, must give the same queue
    k2, d2, c2 = _records(frames)
    h1 = rumination.all_gather_records_async(c2, k2, d2, n_frames)
    h2 = rumination.all_gather_records_async(c2, k2, d2, n_frames)
    for h in (h1, h2):
        gc, gk, gd = h.wait()
        assert torch.equal(gc, counts) and gk.numpy().tobytes() == kp.numpy().tobytes() and torch.equal(gd, desc)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), counts=counts.numpy(), kp=kp.numpy(), desc=desc.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_the_queue():
    for n in (1, 7, 8, 1024, 1000):
        for w in (1, 2, 3, 8):
            b = [rumination.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert rumination.shard_capacity(n, w) == max(h - l for l, h in b)


@pytest.mark.parametrize("n_frames", [4, 5])
def test_two_rank_all_gather_matches_single_process(tmp_path, n_frames):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, n_frames, str(tmp_path)), nprocs=2, join=True)
    kp, desc, counts = _records([synth_frame(300 + i, w=320, h=240) for i in range(n_frames)])
    for r in range(2):
        g = np.load(tmp_path / f"r{r}.npz")
        assert np.array_equal(g["counts"], counts.numpy()), f"rank {r} counts"
        assert g["kp"].tobytes() == kp.numpy().tobytes() and np.array_equal(g["desc"], desc.numpy()), f"rank {r} records"   # kp holds int fields: compare bytes
