"""Golden fixtures of the matchers and optimisers (tests/golden/match_scene*.npz, opt_problems.npz, written by
tests/golden/make_golden_match_opt.py from the oracle): the CPU tests pin the oracle and the scene generators against drift, the GPU
tests check the product against the committed vectors without touching the oracle."""
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def _gen():
    spec = importlib.util.spec_from_file_location("make_golden_match_opt", os.path.join(GOLD, "make_golden_match_opt.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("seed", [0, 3])
def test_oracle_reproduces_matcher_fixture(seed):
    g = np.load(os.path.join(GOLD, f"match_scene{seed}.npz"))
    out = _gen().match_case(seed)
    assert out["inputs_sha256"] == str(g["inputs_sha256"]), "scene generator or extractor oracle changed"
    for k in g.files:
        if k != "inputs_sha256":
            assert np.array_equal(out[k], g[k]), k


def test_oracle_reproduces_optimiser_fixture():
    g = np.load(os.path.join(GOLD, "opt_problems.npz"))
    out = _gen().opt_case()
    for k in g.files:
        if k.endswith("sha256"):
            assert out[k] == str(g[k]), f"{k}: problem generator changed"
        elif g[k].dtype.kind == "f":
            assert np.allclose(out[k], g[k], rtol=1e-6, atol=1e-7), k          # same code, same machine arithmetic: libm differences only
        else:
            assert np.array_equal(out[k], g[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 3])
def test_matchers_against_fixture_on_gpu(seed):
    from rumi_slam_amd.matcher import FeatureVector, FrameView, ORBmatcher
    from scene import K_TUM3, TrackingScene
    g = np.load(os.path.join(GOLD, f"match_scene{seed}.npz"))
    s = TrackingScene(seed)
    F = FrameView(s.cur_keys, s.cur_desc, s.w, s.h, s.sf)
    fm0 = np.full(F.n, -1, np.int32)
    m = ORBmatcher(0.8, True)
    n1, fm1 = m.SearchByProjection_MapPoints(F, s.mappoint_view(), fm0, 3.0)
    assert n1 == int(g["m1_n"]) and np.array_equal(fm1, g["m1_frame_mp"])
    n2, fm2 = m.SearchByProjection_Frame(F, s.Tcw7, K_TUM3, s.last_keys, s.last_mp, s.last_outlier, s.mp_pos, s.mp_desc, s.mp_obs, fm0, 15.0)
    assert n2 == int(g["m2_n"]) and np.array_equal(fm2, g["m2_cur_mp"])
    fv1, fv2 = s.feature_vectors()
    KF = FrameView(s.last_keys, s.last_desc, s.w, s.h, s.sf)
    n3, fm3 = m.SearchByBoW(KF, FeatureVector(fv1), s.last_mp, np.zeros(len(s.mp_obs), np.uint8), F, FeatureVector(fv2))
    assert n3 == int(g["m3_n"]) and np.array_equal(fm3, g["m3_matches"])
    pm0 = np.stack([s.last_keys["x"], s.last_keys["y"]], 1).astype(np.float32)
    n4, m12, _ = ORBmatcher(0.9, True).SearchForInitialization(KF, F, pm0, 100)
    assert n4 == int(g["init_n"]) and np.array_equal(m12, g["init_matches12"])


@pytest.mark.gpu
def test_optimisers_against_fixture_on_gpu():
    from ba_scene import ba_problem, pose_problem
    from rumi_slam_amd.optimizer import Optimizer
    from sim3_scene import sim3_cloud_problem, sim3_pair_problem
    g = np.load(os.path.join(GOLD, "opt_problems.npz"))
    opt = Optimizer()
    close = lambda a, b: np.abs(np.asarray(a, np.float64) - b).max() <= 1e-4 * max(1.0, np.abs(b).max())
    p = pose_problem(41, 300, 0.1)
    ng, T, out = opt.PoseOptimization(p["Xw"], p["obs"], p["inv_sigma2"], p["K"], p["T0"])
    assert ng == int(g["pose_n_good"]) and np.array_equal(out, g["pose_outlier"]) and close(T, g["pose_T"])
    b = ba_problem(seed=42, n_opt=6, n_fixed=2, n_points=400)
    a = (b["kf_pose"], b["kf_fixed"], b["mp_pos"], b["e_mp"], b["e_kf"], b["e_obs"], b["e_w"], b["K"])
    stats, kp, mp, er = opt.LocalBundleAdjustment(*a)
    assert stats[0] == int(g["lba_iterations"]) and np.array_equal(er, g["lba_erase"]) and close(kp, g["lba_kf"]) and close(mp, g["lba_mp"])
    stats, kp, mp = opt.BundleAdjustment(*a, n_iterations=10, robust=True)
    assert stats[0] == int(g["gba_iterations"]) and close(kp, g["gba_kf"]) and close(mp, g["gba_mp"])
    s = sim3_pair_problem(seed=43, n=150)
    nin, nbad, early, S, st = opt.OptimizeSim3(s["S0"], s["P1c"], s["P2c"], s["obs1"], s["obs2"], s["w1"], s["w2"], s["K"], s["K"], 10.0, False, True)
    assert [nin, nbad, int(early)] == g["sim3_counts"].tolist() and np.array_equal(st, g["sim3_status"]) and close(S, g["sim3_S"])
    c = sim3_cloud_problem(seed=44, n_pairs=4, per_pair=80)
    cn, cb, ce, cS, cst = opt.OptimizeSim3(c["S0"], c["P1c"], c["P2c"], c["obs1"], c["obs2"], c["w1"], c["w2"], c["K"], c["K"], 10.0, True, False, c["pair_of"],
                                           c["S_c1w"], c["S_c2w"], c["skip12"], c["skip21"])
    assert [cn, cb, int(ce)] == g["cloud_counts"].tolist() and np.array_equal(cst, g["cloud_status"]) and close(cS, g["cloud_S"])
