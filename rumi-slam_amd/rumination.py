"""Multi-GPU batch extraction of the rumination queue.

The reference queues the frames tracking could not use (CloudImageSampler.cc:46-54, time-sorted at :162-170) and ships them
off-node; here the queue is sharded over the GPUs of one node — frames are independent units of ``ORBextractor::operator()`` —
contiguous block ``[g*F/G, (g+1)*F/G)`` per rank so the gathered result is already time-ordered, followed by ONE exchange step:
an all-gather of fixed-capacity per-frame records (counts, key-points, descriptors) over RCCL / xGMI
(``torch.distributed`` backend "nccl"; "gloo" in the CPU tests).  No other collective is on the path.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_frames, rank, world):
    """Contiguous block of the time-ordered queue owned by `rank`."""
    return (rank * n_frames) // world, ((rank + 1) * n_frames) // world


def shard_capacity(n_frames, world):
    """Largest block any rank owns (blocks are padded to this so the all-gather is one fixed-shape call)."""
    return max(shard_bounds(n_frames, r, world)[1] - shard_bounds(n_frames, r, world)[0] for r in range(world))


class GatheredRecords:
    """An all-gather of one block's records in flight: ``wait()`` makes the current stream wait for it and returns the three
    tensors of the whole queue.  Between the launch and ``wait()`` the caller's stream is free to run the next block's extraction
    (RCCL works on its own stream), which is how ``bench.py`` hides the exchange step behind the compute of the next step."""

    def __init__(self, outs, works, n_frames, world, per):
        self._outs, self._works, self._n, self._world, self._per = outs, works, n_frames, world, per

    def wait(self):
        for w in self._works:
            w.wait()
        self._works = []
        outs, world, per = self._outs, self._world, self._per
        if world == 1:
            return outs
        if self._n % world == 0:                    # even split: the gathered buffers already ARE the queue, in order
            return tuple(g.view((world * per,) + tuple(g.shape[2:])) for g in outs)
        keep = []
        for r in range(world):
            lo, hi = shard_bounds(self._n, r, world)
            keep.append((r, hi - lo))
        cat = lambda g: torch.cat([g[r, :n] for r, n in keep], 0)
        return cat(outs[0]), cat(outs[1]), cat(outs[2])


def all_gather_records_async(counts, kp, desc, n_frames, group=None):
    """Launches the exchange step for THIS rank's block (counts [b,2] i32, kp [b,cap,7] f32, desc [b,cap,32] u8) and returns a
    GatheredRecords handle."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return GatheredRecords((counts, kp, desc), [], n_frames, 1, counts.shape[0])
    per = shard_capacity(n_frames, world)
    b = counts.shape[0]

    def pad(t):
        if b == per:
            return t.contiguous()
        out = torch.zeros((per,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        out[:b] = t
        return out

    outs, works = [], []
    for t in (counts, kp, desc):
        t = pad(t)
        g = torch.empty((world * per,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)   # concatenated form: accepted by RCCL and gloo
        works.append(dist.all_gather_into_tensor(g, t, group=group, async_op=True))
        outs.append(g.view((world, per) + tuple(t.shape[1:])))
    return GatheredRecords(outs, works, n_frames, world, per)


def all_gather_records(counts, kp, desc, n_frames, group=None):
    """counts [b,2] i32, kp [b,cap,7] f32, desc [b,cap,32] u8 of THIS rank's block -> the same three tensors for all
    n_frames frames, in queue order, on every rank."""
    return all_gather_records_async(counts, kp, desc, n_frames, group).wait()


def extract_queue(extract_fn, frames_of_rank, n_frames, group=None):
    """extract_fn(frames) -> (kp [b,cap,7], desc [b,cap,32], counts [b,2]) on this rank's block; returns the gathered queue."""
    kp, desc, counts = extract_fn(frames_of_rank)
    return all_gather_records(counts, kp, desc, n_frames, group)
