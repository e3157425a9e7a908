"""Builds tests/cpp/test_facade.cc (C++ facade classes with the reference's signatures + a mock data model) against
librumi_hip.so and the oracle, and runs it on the GPU."""
import os
import subprocess

import pytest

from rumi_slam_amd.synth import synth_frame, warp_frame

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_facade_test(out, source="test_facade.cc", extra_inc=()):
    fac = os.path.join(ROOT, "rumi_slam_amd", "facade")
    inc = []
    for d in extra_inc:
        inc += ["-I", d]
    cmd = ["g++", "-O1", "-std=c++17", "-ffp-contract=off"] + inc + ["-I", fac, "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "oracle"),
           "-I", os.path.join(ROOT, "tests", "cpp"), os.path.join(ROOT, "tests", "cpp", source), os.path.join(fac, "ORBextractor.cc"),
           "-L", os.path.join(ROOT, "rumi_slam_amd"), "-lrumi_hip", "-L", os.path.join(ROOT, "oracle"), "-loracle",
           "-Wl,-rpath," + os.path.join(ROOT, "rumi_slam_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-lpthread", "-o", out]
    subprocess.check_call(cmd)


def test_facade_compiles(tmp_path):
    build_facade_test(str(tmp_path / "test_facade"))            # CPU: the facade + mock data model compile and link
    # ... and so do the sections that take Sophus::Sim3f / write back through Sophus::SE3f, against tests/cpp/mock_sophus.h
    build_facade_test(str(tmp_path / "test_facade_sophus"), "test_facade_sophus.cc")
    # ... and the non-template shells (facade/shells/*.cc) against the reference's own class declarations (tests/cpp/ref_decls: signatures only)
    build_facade_test(str(tmp_path / "test_shells"), "test_shells.cc", extra_inc=[os.path.join(ROOT, "tests", "cpp", "ref_decls")])


@pytest.mark.gpu
def test_facade_against_oracle(tmp_path):
    exe = str(tmp_path / "test_facade")
    build_facade_test(exe)
    img0 = synth_frame(4242)
    img1, _ = warp_frame(img0, 17)
    img0.tofile(tmp_path / "f0.bin"); img1.tofile(tmp_path / "f1.bin")
    env = dict(os.environ, RUMI_NO_TORCH="1")
    r = subprocess.run([exe, str(tmp_path / "f0.bin"), str(tmp_path / "f1.bin")], capture_output=True, text=True, env=env, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_facade_sophus_overloads(tmp_path):
    """The RUMI_HAVE_SOPHUS sections of facade/*.h (Sim3 projections, Fuse with a Sim3, SearchBySim3, SearchForTriangulation with the
    fundamental matrix formed by the facade, SetPose write-back) compiled against a minimal Eigen/Sophus mock and run on the GPU."""
    exe = str(tmp_path / "test_facade_sophus")
    build_facade_test(exe, "test_facade_sophus.cc")
    img0 = synth_frame(4242)
    img1, _ = warp_frame(img0, 17)
    img0.tofile(tmp_path / "f0.bin"); img1.tofile(tmp_path / "f1.bin")
    env = dict(os.environ, RUMI_NO_TORCH="1")
    r = subprocess.run([exe, str(tmp_path / "f0.bin"), str(tmp_path / "f1.bin")], capture_output=True, text=True, env=env, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_shells_forward_to_the_templates(tmp_path):
    """facade/shells/ORBmatcher.cc and Optimizer_hot.cc (reference signatures, non-template) give what the templates give."""
    exe = str(tmp_path / "test_shells")
    build_facade_test(exe, "test_shells.cc", extra_inc=[os.path.join(ROOT, "tests", "cpp", "ref_decls")])
    img0 = synth_frame(4242)
    img1, _ = warp_frame(img0, 17)
    img0.tofile(tmp_path / "f0.bin"); img1.tofile(tmp_path / "f1.bin")
    env = dict(os.environ, RUMI_NO_TORCH="1")
    r = subprocess.run([exe, str(tmp_path / "f0.bin"), str(tmp_path / "f1.bin")], capture_output=True, text=True, env=env, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
