"""Consumes tests/golden/ref/*.npy when a maintainer has produced them with tools/dump_reference_vectors.cc inside the reference workspace (real
OpenCV 3.4, Eigen, g2o, ORBextractor).  Without those files every test here is skipped and the oracle stays "parity unpinned" for the OpenCV- and
Eigen-owned steps; with them the oracle -- and through it the HIP kernels, held bit-exact to the oracle by the -m gpu suite
-- is pinned against the reference binary itself."""
import glob
import os

import numpy as np
import pytest

import oracle_lib as O
from golden.make_ref_inputs import FRAMES, POSES
from ba_scene import pose_problem
from rumi_slam_amd.synth import synth_frame

REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref")
have = os.path.isdir(REF) and bool(glob.glob(os.path.join(REF, "kp_*.npy")))
pytestmark = pytest.mark.skipif(not have, reason="tests/golden/ref/ not populated (run tools/dump_reference_vectors.cc in the reference workspace)")


def _ref(name):
    return np.load(os.path.join(REF, name + ".npy"))


def _variant():
    """Which GaussianBlur the reference's OpenCV has: decided on the first frame's level 0, then required everywhere."""
    img = synth_frame(**FRAMES[0])
    for v in (0, 1):
        if np.array_equal(O.gaussian_blur(img, v), _ref("blur_0_0")):
            return v
    pytest.fail("neither documented GaussianBlur variant reproduces the reference's blur_0_0.npy: add the variant of this OpenCV build to the oracle and to k_blur")


@pytest.mark.parametrize("f", range(len(FRAMES)))
def test_pyramid_blur_fast_and_records(f):
    v = _variant()
    img = synth_frame(**FRAMES[f])
    orc = O.OracleExtractor(1000, 1.2, 8, 20, 7, blur_variant=v)
    mono, kp, desc = orc.extract(img, (0, 1000))
    for l in range(8):
        lv = _ref(f"level_{f}_{l}")
        assert np.array_equal(orc.level(l), lv), f"cv::resize chain, frame {f} level {l}"
        assert np.array_equal(O.gaussian_blur(lv, v), _ref(f"blur_{f}_{l}")), f"cv::GaussianBlur variant {v}, frame {f} level {l}"
        for thr in (20, 7):
            got = O.fast_cell(lv, thr)
            ref = _ref(f"fast{thr}_{f}_{l}")
            assert len(got) == len(ref) and np.array_equal(np.stack([got["x"], got["y"], got["response"]], 1), ref), f"cv::FAST thr {thr}, frame {f} level {l}"
    assert mono == int(_ref(f"mono_{f}")[0])
    assert kp.view(np.float32).reshape(-1, 7).tobytes() == _ref(f"kp_{f}").tobytes(), "key-point records"
    assert np.array_equal(desc, _ref(f"desc_{f}")), "descriptors"


def test_fast_atan2_grid():
    ref = _ref("atan2")
    got = np.array([[O.fast_atan2(y * 37.0, x * 53.0) for x in range(-40, 41)] for y in range(-40, 41)], np.float32)
    assert got.tobytes() == ref.tobytes()


@pytest.mark.parametrize("p", range(len(POSES)))
def test_pose_optimization_against_g2o(p):
    pr = pose_problem(POSES[p], 300, 0.1)
    n_good, T, _ = O.pose_optimization(pr["Xw"], pr["obs"], pr["inv_sigma2"], pr["K"], pr["T0"])
    ref = _ref(f"pose_{p}")
    assert n_good == int(ref[0])
    assert np.allclose(T.astype(np.float64), ref[1:8], rtol=1e-4, atol=1e-6)        # north_star tolerance: 1e-4 relative
