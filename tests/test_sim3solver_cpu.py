"""CPU checks of the Sim3Solver path's test infrastructure: the glibc rand() restatement against the real C library, the minimal-set
draws, and the oracle's Horn solver against the transform a synthetic problem was built with (an independent property: upstream's
Eigen::EigenSolver is not in the tree, so the closed form is "parity unpinned" and anchored on what it must compute)."""
import numpy as np
import pytest

import oracle_lib as O
from rumi_slam_amd.sim3solver import GlibcRand, Sim3Solver
from sim3_scene import sim3_ransac_problem


@pytest.mark.parametrize("seed", [0, 1, 42, 123456789, 2 ** 31 + 5])
def test_glibc_rand_restatement_matches_libc(seed):
    O.libc_srand(seed)
    ref = [O.libc_rand() for _ in range(2000)]
    g = GlibcRand(seed)
    assert [g.rand() for _ in range(2000)] == ref


def test_minimal_sets_follow_the_reference_draw_order():
    n, H = 57, 300
    tri = O.sim3_draw_triples(0, n, H)
    s = Sim3Solver(None, np.ones((n, 3)), np.ones((n, 3)), np.ones(n), np.ones(n), np.ones(4), np.ones(4), rng=GlibcRand(0))
    mine, _ = s._draw_block(H)
    assert np.array_equal(mine, tri)
    assert all(len(set(t)) == 3 for t in tri.tolist())


def test_ransac_parameters():
    s = Sim3Solver(None, np.ones((80, 3)), np.ones((80, 3)), np.ones(80), np.ones(80), np.ones(4), np.ones(4))
    s.SetRansacParameters(0.99, 20, 300)
    eps = np.float32(20) / np.float32(80)
    assert s.mRansacMaxIts == min(300, int(np.ceil(np.log(1 - 0.99) / np.log(1 - float(eps) ** 3))))
    s.SetRansacParameters(0.99, 80, 300)
    assert s.mRansacMaxIts == 1
    s.SetRansacParameters(0.99, 6, 5)
    assert s.mRansacMaxIts == 5


@pytest.mark.parametrize("fix_scale", [False, True])
def test_oracle_closed_form_recovers_the_similarity(fix_scale):
    pr = sim3_ransac_problem(3, scale=1.0 if fix_scale else 1.4, outlier_frac=0.0)
    n = len(pr["X1"])
    tri = O.sim3_draw_triples(5, n, 30)
    r = O.sim3_ransac(pr["X1"], pr["X2"], pr["sigma2_1"], pr["sigma2_2"], pr["K"], pr["K"], tri, fix_scale=fix_scale, score=pr["score"])
    assert r["valid"].all()
    # every clean minimal set maps its own three points of camera 2 onto camera 1
    for h in range(len(tri)):
        p2, p1 = pr["X2"][tri[h]].astype(np.float64), pr["X1"][tri[h]].astype(np.float64)
        back = r["s"][h] * (p2 @ r["R"][h].astype(np.float64).T) + r["t"][h]
        assert np.abs(back - p1).max() < 0.03
        assert abs(np.linalg.det(r["R"][h].astype(np.float64)) - 1) < 1e-5
    best = int(np.argmax(r["n_inliers"]))
    assert r["n_inliers"][best] >= n - 2
    assert np.abs(r["R"][best] - pr["R12"]).max() < 0.02 and np.abs(r["t"][best] - pr["t12"]).max() < 0.1
    assert abs(r["s"][best] - pr["s12"]) < 0.03
    assert r["median"][best] > 0.6


def test_oracle_flags_the_degenerate_minimal_set():
    pr = sim3_ransac_problem(4)
    X1, X2 = pr["X1"].copy(), pr["X2"].copy()
    X1[:3] = X1[0]; X2[:3] = X2[0]
    r = O.sim3_ransac(X1, X2, pr["sigma2_1"], pr["sigma2_2"], pr["K"], pr["K"], np.array([[0, 1, 2], [5, 9, 30]], np.int32))
    assert not r["valid"][0] and r["valid"][1]
