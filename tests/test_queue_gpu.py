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
