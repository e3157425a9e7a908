"""The rumination queue on N devices from ONE process behind the C ABI (include/rumi_queue.h): a C++ host program through the header alone, and the
Python mirror.  One GPU here: 1, 2 and 3 LOGICAL shards aliased to device 0 (device-to-device copies stand in for the collective when ordinals
repeat); a single shard with its own device goes through RCCL's ncclAllGather, so the run-time binding of librccl is exercised too."""
import os
import subprocess

import numpy as np
import pytest

from rumi_slam_amd.synth import synth_frame

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_queue_test(out):
    cmd = ["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_queue.cc"),
           "-L", os.path.join(ROOT, "rumi_slam_amd"), "-lrumi_hip", "-Wl,-rpath," + os.path.join(ROOT, "rumi_slam_amd"), "-L", "/opt/rocm/lib", "-lamdhip64", "-lpthread", "-o", out]      # (HIP only for the test's own reference buffer)
    subprocess.check_call(cmd)


def test_queue_host_program_compiles(tmp_path):
    build_queue_test(str(tmp_path / "test_queue"))


@pytest.mark.gpu
def test_queue_through_the_header(tmp_path):
    exe = str(tmp_path / "test_queue")
    build_queue_test(exe)
    F = 14
    np.stack([synth_frame(900 + i) for i in range(F)]).tofile(tmp_path / "frames.bin")
    r = subprocess.run([exe, str(F), "640", "480", str(tmp_path / "frames.bin")], capture_output=True, text=True, env=dict(os.environ, RUMI_NO_TORCH="1"), timeout=600)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "queue OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("shards", [1, 2, 3])
def test_queue_mirror_equals_the_single_extractor(shards):
    import oracle_lib as O
    from rumi_slam_amd.queue import RuminationQueue, split_records
    F = 7
    frames = [synth_frame(1200 + i) for i in range(F)]
    q = RuminationQueue(1000, 1.2, 8, 20, 7, [0] * shards, max_block=(F + shards - 1) // shards)
    rec, dptr = q.extract(frames)
    assert len(dptr) == shards and all(dptr) and q.uses_rccl == (shards == 1)
    counts, kp, desc = split_records(rec, q.cap)
    orc = O.OracleExtractor(1000, 1.2, 8, 20, 7)
    for i in (0, F // 2, F - 1):
        mono, okp, odesc = orc.extract(frames[i], (0, 1000))
        n = int(counts[i, 0])
        assert n == len(okp) and int(counts[i, 1]) == mono
        assert kp[i, :n].tobytes() == okp.tobytes() and np.array_equal(desc[i, :n], odesc)
    assert [q.row(F, f) for f in range(F)] == sorted(q.row(F, f) for f in range(F))
    print(q.last_ms())


@pytest.mark.gpu
@pytest.mark.parametrize("shards", [1, 2])
def test_queue_records_reach_pinned_and_pageable_memory_alike(shards):
    """The records go back from every shard's own device: into pinned memory sub-chunk by sub-chunk behind the kernels, into pageable memory in one copy per
    shard.  Both must hold the same bytes as the gathered block on the device (a queue of several sub-chunks per shard, uneven last block)."""
    import torch
    from rumi_slam_amd.queue import RuminationQueue
    F = 150
    base = [synth_frame(1500 + i) for i in range(10)]
    frames = [base[i % 10] if i < 10 else np.roll(base[i % 10], (3 * (i // 10), 5 * (i // 10)), (0, 1)) for i in range(F)]
    q = RuminationQueue(1000, 1.2, 8, 20, 7, [0] * shards, max_block=(F + shards - 1) // shards)
    pageable = np.zeros((F, q.record_bytes), np.uint8)
    pinned = torch.zeros((F, q.record_bytes), dtype=torch.uint8).pin_memory().numpy()
    _, dptr = q.extract(frames, out=pageable)
    q.extract(frames, out=pinned)
    assert np.array_equal(pageable, pinned)
    counts = pageable[:, :8].copy().view(np.int32).reshape(F, 2)
    assert counts[:, 0].min() > 500
    # the gathered block of every shard (device memory): frame f at row q.row(F, f)
    import ctypes as C
    from rumi_slam_amd import capi
    hip = C.CDLL("libamdhip64.so")
    for g in range(shards):
        blk = np.zeros((shards * q.block_capacity, q.record_bytes), np.uint8)
        assert hip.hipMemcpy(C.c_void_p(blk.ctypes.data), C.c_void_p(dptr[g]), C.c_size_t(blk.nbytes), 2) == 0      # hipMemcpyDeviceToHost
        for f in (0, 1, F // 2, F - 2, F - 1):
            r = q.row(F, f)
            n = int(counts[f, 0])
            assert np.array_equal(blk[r, :8 + 28 * n], pinned[f, :8 + 28 * n])
            assert np.array_equal(blk[r, 8 + 28 * q.cap:8 + 28 * q.cap + 32 * n], pinned[f, 8 + 28 * q.cap:8 + 28 * q.cap + 32 * n])
