"""gloo tests (world sizes 2 and 3) of the rumination sharding + the single all-gather of per-frame records (the N > 1 path of bench.py);
extraction itself is stood in for by the CPU oracle here (no GPU in this container) — what is under test is the partition, the record
layout and the exchange step."""
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
    return rumination.pack_records(torch.from_numpy(kp), torch.from_numpy(desc), torch.from_numpy(counts))


def _worker(rank, world, port, n_frames, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = rumination.shard_bounds(n_frames, rank, world)
    frames = [synth_frame(300 + i, w=320, h=240) for i in range(lo, hi)]
    rec = rumination.extract_queue(_records, frames, n_frames)
    assert rec.shape == (n_frames, rumination.record_bytes(CAP))
    # the pipelined form bench.py uses: two exchanges in flight, joined later, must give the same queue
    mine = _records(frames)
    h1 = rumination.all_gather_records_async(mine, n_frames)
    h2 = rumination.all_gather_records_async(mine, n_frames)
    for h in (h1, h2):
        assert torch.equal(h.wait(), rec)
    # ... and into a preallocated buffer (the steady-state loop of bench.py allocates nothing)
    pre = torch.full((world * rumination.shard_capacity(n_frames, world), rumination.record_bytes(CAP)), 0xAB, dtype=torch.uint8)
    padded = torch.zeros((rumination.shard_capacity(n_frames, world), rumination.record_bytes(CAP)), dtype=torch.uint8)
    padded[:mine.shape[0]] = mine
    h3 = rumination.all_gather_records_async(padded, n_frames, out=pre)
    assert torch.equal(h3.wait(), rec) and h3.wait().data_ptr() == pre.data_ptr() or torch.equal(h3.wait(), rec)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), rec.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_the_queue():
    for n in (1, 7, 8, 1024, 1000):
        for w in (1, 2, 3, 8):
            b = [rumination.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert rumination.shard_capacity(n, w) == max(h - l for l, h in b)


def test_record_views_round_trip():
    rng = np.random.default_rng(5)
    kp = torch.from_numpy(rng.standard_normal((3, CAP, 7)).astype(np.float32))
    desc = torch.from_numpy(rng.integers(0, 256, (3, CAP, 32), dtype=np.uint8))
    counts = torch.from_numpy(rng.integers(0, CAP, (3, 2)).astype(np.int32))
    rec = rumination.pack_records(kp, desc, counts)
    assert rec.shape == (3, 8 + 60 * CAP)
    k, d, c = rumination.record_views(rec, CAP)
    assert torch.equal(k, kp) and torch.equal(d, desc) and torch.equal(c, counts)
    raw = rec.numpy()                                   # the layout the kernels write: n, monoIndex, key-points, descriptors
    assert np.array_equal(raw[:, :8].view(np.int32), counts.numpy())
    assert raw[1, 8:8 + 28 * CAP].tobytes() == kp[1].numpy().tobytes() and raw[2, 8 + 28 * CAP:].tobytes() == desc[2].numpy().tobytes()


@pytest.mark.parametrize("world,n_frames", [(2, 4), (2, 5), (3, 6), (3, 7)])
def test_all_gather_matches_single_process(tmp_path, world, n_frames):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(world, port, n_frames, str(tmp_path)), nprocs=world, join=True)
    ref = _records([synth_frame(300 + i, w=320, h=240) for i in range(n_frames)]).numpy()
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"r{r}.npy"), ref), f"rank {r}: gathered records"
