"""Multi-GPU batch extraction of the rumination queue.

The reference queues the frames tracking could not use (CloudImageSampler.cc:46-54, time-sorted at :162-170) and ships them
off-node; here the queue is sharded over the GPUs of one node — frames are independent units of ``ORBextractor::operator()`` —
contiguous block ``[g*F/G, (g+1)*F/G)`` per rank so the gathered result is already time-ordered, followed by ONE exchange step:
a single all-gather of fixed-capacity per-frame records over RCCL / xGMI (``torch.distributed`` backend "nccl"; "gloo" in the CPU
tests).  No other collective is on the path.

Record of one frame (``record_bytes(cap)`` = 8 + 60 * cap bytes, written as such by ``rumi_orb_extract_batch_records_async``):
    int32 n; int32 monoIndex; RumiKeyPoint kp[cap] (28 B each); uint8 desc[cap][32]
"""
import torch
import torch.distributed as dist


def record_bytes(cap):
    return 8 + 60 * int(cap)


def record_views(records, cap):
    """(kp [F,cap,7] f32, desc [F,cap,32] u8, counts [F,2] i32) as VIEWS of a [F, record_bytes(cap)] u8 record tensor (no copy)."""
    F, rb = records.shape
    assert rb == record_bytes(cap) and records.is_contiguous()
    i32 = records.view(torch.int32)                      # [F, rb / 4]
    f32 = records.view(torch.float32)
    counts = i32[:, :2]
    kp = torch.as_strided(f32, (F, cap, 7), (rb // 4, 7, 1), f32.storage_offset() + 2)
    desc = torch.as_strided(records, (F, cap, 32), (rb, 32, 1), records.storage_offset() + 8 + 28 * cap)
    return kp, desc, counts


def pack_records(kp, desc, counts):
    """The three arrays of a block -> its [b, record_bytes(cap)] u8 record tensor (used where the extractor did not write records itself)."""
    b, cap = kp.shape[0], kp.shape[1]
    rec = torch.zeros((b, record_bytes(cap)), dtype=torch.uint8, device=kp.device)
    k, d, c = record_views(rec, cap)
    k.copy_(kp); d.copy_(desc); c.copy_(counts)
    return rec


def shard_bounds(n_frames, rank, world):
    """Contiguous block of the time-ordered queue owned by `rank`."""
    return (rank * n_frames) // world, ((rank + 1) * n_frames) // world


def shard_capacity(n_frames, world):
    """Largest block any rank owns (blocks are padded to this so the all-gather is one fixed-shape call)."""
    return max(shard_bounds(n_frames, r, world)[1] - shard_bounds(n_frames, r, world)[0] for r in range(world))


class GatheredRecords:
    """The all-gather of one block's records in flight: ``wait()`` makes the current stream wait for it and returns the record tensor
    of the whole queue, [n_frames, record_bytes], in queue order.  Between the launch and ``wait()`` the caller's stream is free to
    run the next block's extraction (RCCL works on its own stream), which is how ``bench.py`` hides the exchange behind the next step."""

    def __init__(self, gathered, work, n_frames, world, per):
        self._g, self._work, self._n, self._world, self._per = gathered, work, n_frames, world, per

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        if self._world == 1 or self._n % self._world == 0:      # even split: the gathered buffer already IS the queue, in order
            return self._g[:self._n]
        g = self._g.view(self._world, self._per, -1)            # uneven split: drop each rank's padding records
        return torch.cat([g[r, :shard_bounds(self._n, r, self._world)[1] - shard_bounds(self._n, r, self._world)[0]]
                          for r in range(self._world)], 0)


def all_gather_records_async(records, n_frames, group=None, out=None):
    """Launches THE exchange step for this rank's block of records ([b, record_bytes] u8; b <= shard_capacity) — one
    ``all_gather_into_tensor`` — and returns a GatheredRecords handle.  A block shorter than the capacity (uneven split) is padded
    with empty records; pass an already padded [shard_capacity, record_bytes] tensor to avoid the copy.  out: a preallocated
    [world * shard_capacity, record_bytes] u8 tensor for the gathered queue (a steady-state loop allocates nothing)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return GatheredRecords(records, None, n_frames, 1, records.shape[0])
    per = shard_capacity(n_frames, world)
    if records.shape[0] != per:
        padded = torch.zeros((per, records.shape[1]), dtype=records.dtype, device=records.device)
        padded[:records.shape[0]] = records
        records = padded
    records = records.contiguous()
    g = out if out is not None else torch.empty((world * per, records.shape[1]), dtype=records.dtype, device=records.device)
    assert g.shape == (world * per, records.shape[1]) and g.is_contiguous()
    work = dist.all_gather_into_tensor(g, records, group=group, async_op=True)
    return GatheredRecords(g, work, n_frames, world, per)


def all_gather_records(records, n_frames, group=None):
    """This rank's block of records -> the records of all n_frames frames, in queue order, on every rank."""
    return all_gather_records_async(records, n_frames, group).wait()


def extract_queue(extract_fn, frames_of_rank, n_frames, group=None):
    """extract_fn(frames) -> [b, record_bytes] u8 records of this rank's block; returns the gathered queue's records."""
    return all_gather_records(extract_fn(frames_of_rank), n_frames, group)
