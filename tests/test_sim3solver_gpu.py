"""GPU parity of rumi_sim3_ransac (Sim3Solver::iterate, R/lib_src/Sim3Solver.cc:159-404) against the CPU oracle, and of the host mirror's
RANSAC state machine against a plain replay of upstream's loop over the oracle's per-iteration results.
Tolerance: the hypotheses are float closed-form solutions around an eigen-decomposition (Jacobi in double on both sides, different sweeps;
device sinf/cosf/atan2 vs glibc): R, t, s within 1e-4 relative; inlier sets and ratios are integer work and must be equal on these scenes
(their thresholds sit far from the re-projection errors of inliers and of gross outliers)."""
import numpy as np
import pytest

import oracle_lib as O
from rumi_slam_amd.capi import RumiError
from rumi_slam_amd.sim3solver import GlibcRand, Sim3Solver
from sim3_scene import sim3_ransac_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def opt():
    from rumi_slam_amd.optimizer import Optimizer
    return Optimizer()


def _close(g, o):
    assert np.array_equal(g["valid"], o["valid"])
    v = o["valid"]
    assert np.abs(g["R"][v] - o["R"][v]).max() < 1e-4
    assert np.abs(g["t"][v] - o["t"][v]).max() < 1e-4 * max(1.0, np.abs(o["t"][v]).max())
    assert np.abs(g["s"][v] - o["s"][v]).max() < 1e-4 * max(1.0, np.abs(o["s"][v]).max())


@pytest.mark.parametrize("seed,fix_scale", [(0, False), (1, False), (2, True)])
def test_hypotheses_match_oracle(opt, seed, fix_scale):
    pr = sim3_ransac_problem(seed, scale=1.0 if fix_scale else 1.25)
    n = len(pr["X1"])
    tri = O.sim3_draw_triples(seed, n, 300)
    args = (pr["X1"], pr["X2"], pr["sigma2_1"], pr["sigma2_2"], pr["K"], pr["K"], tri)
    o = O.sim3_ransac(*args, fix_scale=fix_scale, score=pr["score"])
    g = opt.Sim3Ransac(*args, fix_scale=fix_scale, score=pr["score"])
    _close(g, o)
    assert np.array_equal(g["n_inliers"], o["n_inliers"])
    assert np.array_equal(g["inliers"], o["inliers"])
    assert np.array_equal(g["ratio"], o["ratio"])
    assert np.array_equal(g["median"], o["median"])
    assert o["n_inliers"].max() >= (~pr["bad"]).sum() - 2
    # without the score set, and without the inlier masks
    g2 = opt.Sim3Ransac(*args, fix_scale=fix_scale, want_inliers=False)
    assert g2["inliers"] is None and g2["median"] is None and np.array_equal(g2["n_inliers"], o["n_inliers"])


def _replay(o, min_inliers, max_its, n_per_call, mode, best_ratio=0.0):
    """upstream's loop (:176-216 / :243-289 / :315-403) over per-iteration results: (iteration it returns at or None, iterations run, best inliers, best ratio)."""
    it, best_n = 0, 0
    while it < max_its:
        for _ in range(n_per_call):
            if it >= max_its:
                break
            n_i = int(o["n_inliers"][it]); r_i = float(o["median"][it])
            it += 1
            ok = n_i >= best_n if mode != "rumination" else (r_i >= best_ratio and n_i >= best_n)
            if ok:
                best_n = n_i
                if mode == "rumination":
                    best_ratio = r_i
                if (n_i > min_inliers) if mode != "rumination" else (r_i > 0.10 and n_i > min_inliers):
                    return it - 1, it, best_n, best_ratio
    return None, it, best_n, best_ratio


@pytest.mark.parametrize("mode", ["plain", "converge", "rumination"])
def test_iterate_state_machine(opt, mode):
    pr = sim3_ransac_problem(7, outlier_frac=0.45)
    n = len(pr["X1"])
    min_inl = 40
    s = Sim3Solver(opt, pr["X1"], pr["X2"], pr["sigma2_1"], pr["sigma2_2"], pr["K"], pr["K"], rng=GlibcRand(0), indices1=np.arange(n) * 2, mN1=2 * n)
    s.SetRansacParameters(0.99, min_inl, 300)
    tri = O.sim3_draw_triples(0, n, s.mRansacMaxIts)
    o = O.sim3_ransac(pr["X1"], pr["X2"], pr["sigma2_1"], pr["sigma2_2"], pr["K"], pr["K"], tri, score=pr["score"])
    want_it, want_total, want_best, want_ratio = _replay(o, min_inl, s.mRansacMaxIts, 20, mode)
    assert want_it is not None and want_it >= 1, "scene must need more than one iteration"
    conv, no_more, ratio, calls = False, False, 0.0, 0
    while not conv and not no_more:
        calls += 1
        if mode == "plain":
            T, no_more, vb, n_in = s.iterate(20)
            conv = n_in > 0
        elif mode == "converge":
            T, no_more, vb, n_in, conv = s.iterate_converge(20)
        else:
            T, no_more, conv, ratio = s.iterate_rumination(20, pr["score"], ratio)
    assert conv and s.mnIterations == want_it + 1 and s.mnBestInliers == want_best
    assert calls == want_it // 20 + 1
    if mode != "rumination":
        assert n_in == int(o["n_inliers"][want_it]) and vb.sum() == n_in and not vb[1::2].any()
        assert np.array_equal(vb[0::2], o["inliers"][want_it])
    else:
        assert ratio == pytest.approx(want_ratio)
    assert np.abs(s.GetEstimatedRotation() - o["R"][want_it]).max() < 1e-4 and abs(s.GetEstimatedScale() - o["s"][want_it]) < 1e-4
    assert np.abs(T[:3, 3] - o["t"][want_it]).max() < 1e-3
    # the generator stands where upstream's loop would have left rand(): after the draws of the returning iteration
    g = GlibcRand(0)
    for _ in range(3 * (want_it + 1)):
        g.rand()
    assert s.rng.rand() == g.rand()


def test_too_few_correspondences_and_exhaustion(opt):
    pr = sim3_ransac_problem(9, outlier_frac=0.9, n_solver=40)
    s = Sim3Solver(opt, pr["X1"], pr["X2"], pr["sigma2_1"], pr["sigma2_2"], pr["K"], pr["K"], rng=GlibcRand(0))
    s.SetRansacParameters(0.99, 60, 300)                      # more than there are correspondences (:165-168)
    T, no_more, vb, n_in = s.iterate(20)
    assert no_more and n_in == 0 and np.array_equal(T, np.eye(4, dtype=np.float32))
    s.SetRansacParameters(0.99, 12, 25)                       # nothing reaches 12 inliers: two calls exhaust the 25 iterations
    assert s.mRansacMaxIts == 25
    T, no_more, vb, n_in = s.iterate(20)
    assert not no_more and n_in == 0 and s.mnIterations == 20
    T, no_more, vb, n_in = s.iterate(20)
    assert no_more and s.mnIterations == 25


def test_degenerate_set_and_bad_arguments(opt):
    pr = sim3_ransac_problem(4)
    X1, X2 = pr["X1"].copy(), pr["X2"].copy()
    X1[:3] = X1[0]; X2[:3] = X2[0]
    tri = np.array([[0, 1, 2], [5, 9, 30]], np.int32)
    g = opt.Sim3Ransac(X1, X2, pr["sigma2_1"], pr["sigma2_2"], pr["K"], pr["K"], tri)
    o = O.sim3_ransac(X1, X2, pr["sigma2_1"], pr["sigma2_2"], pr["K"], pr["K"], tri)
    assert not g["valid"][0] and g["valid"][1]
    _close(g, o)
    with pytest.raises(RumiError):
        opt.Sim3Ransac(X1, X2, pr["sigma2_1"], pr["sigma2_2"], pr["K"], pr["K"], np.array([[0, 1, len(X1)]], np.int32))
    with pytest.raises(RumiError):
        opt.Sim3Ransac(X1[:2], X2[:2], pr["sigma2_1"][:2], pr["sigma2_2"][:2], pr["K"], pr["K"], np.array([[0, 1, 1]], np.int32))
    assert len(opt.Sim3Ransac(X1, X2, pr["sigma2_1"], pr["sigma2_2"], pr["K"], pr["K"], np.zeros((0, 3), np.int32))["n_inliers"]) == 0
