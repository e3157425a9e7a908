"""GPU parity of PoseOptimization / LocalBundleAdjustment against the CPU oracle: poses and landmarks within 1e-4
relative (north_star), inlier counts and erase flags identical."""
import time

import numpy as np
import pytest

import oracle_lib as O
from ba_scene import ba_problem, pose_problem

pytestmark = pytest.mark.gpu
RTOL = 1e-4


@pytest.fixture(scope="module")
def opt():
    from rumi_slam_amd.optimizer import Optimizer
    return Optimizer()


def _pose_close(a, b, tag=""):
    # quaternion (unit, w >= 0 by construction) and translation: 1e-4 relative to the vector norms
    assert np.linalg.norm(a[:4] - b[:4]) <= RTOL * max(1.0, np.linalg.norm(b[:4])), f"{tag} rotation {a[:4]} vs {b[:4]}"
    assert np.linalg.norm(a[4:] - b[4:]) <= RTOL * max(1e-2, np.linalg.norm(b[4:])), f"{tag} translation {a[4:]} vs {b[4:]}"


@pytest.mark.parametrize("seed,n,frac", [(0, 300, 0.1), (1, 600, 0.2), (2, 100, 0.0), (3, 1500, 0.3), (4, 12, 0.1), (5, 9, 0.0)])
def test_pose_optimization(opt, seed, n, frac):
    p = pose_problem(seed, n, frac)
    ng_ref, T_ref, out_ref = O.pose_optimization(p["Xw"], p["obs"], p["inv_sigma2"], p["K"], p["T0"])
    ng, T, out = opt.PoseOptimization(p["Xw"], p["obs"], p["inv_sigma2"], p["K"], p["T0"])
    assert ng == ng_ref
    assert np.array_equal(out, out_ref), f"{np.count_nonzero(out != out_ref)} outlier flags differ"
    _pose_close(T, T_ref, f"seed {seed}")


def test_pose_optimization_degenerate(opt):
    p = pose_problem(7, 300)
    ng, T, out = opt.PoseOptimization(p["Xw"][:2], p["obs"][:2], p["inv_sigma2"][:2], p["K"], p["T0"])
    assert ng == 0 and np.array_equal(T, p["T0"]) and not out.any()          # < 3 correspondences: returns 0, pose untouched
    ng, T, out = opt.PoseOptimization(p["Xw"][:0], p["obs"][:0], p["inv_sigma2"][:0], p["K"], p["T0"])
    assert ng == 0


def test_pose_optimization_batch(opt):
    probs = [pose_problem(20 + i, 150 + 37 * i, 0.15) for i in range(7)]
    start = np.cumsum([0] + [len(p["inv_sigma2"]) for p in probs]).astype(np.int32)
    ng, T, out = opt.PoseOptimizationBatch(start, np.concatenate([p["Xw"] for p in probs]), np.concatenate([p["obs"] for p in probs]),
                                           np.concatenate([p["inv_sigma2"] for p in probs]), probs[0]["K"], np.stack([p["T0"] for p in probs]))
    for i, p in enumerate(probs):
        ng_ref, T_ref, out_ref = O.pose_optimization(p["Xw"], p["obs"], p["inv_sigma2"], p["K"], p["T0"])
        assert ng[i] == ng_ref and np.array_equal(out[start[i]:start[i + 1]], out_ref)
        _pose_close(T[i], T_ref, f"batch {i}")


@pytest.mark.parametrize("cfg", [dict(seed=0, n_opt=20, n_fixed=5, n_points=3000), dict(seed=1, n_opt=6, n_fixed=2, n_points=500),
                                 dict(seed=2, n_opt=1, n_fixed=3, n_points=200), dict(seed=3, n_opt=12, n_fixed=1, n_points=1500, outlier_frac=0.15),
                                 # the tile solver's shapes: unknowns a multiple of 16 (the right-hand side alone in the last tile row: 8, 16, 24
                                 # key-frames), a partial last panel (3, 27), the largest window it takes (29) and the first one it does not (30)
                                 dict(seed=11, n_opt=8, n_fixed=2, n_points=400), dict(seed=12, n_opt=16, n_fixed=2, n_points=600),
                                 dict(seed=13, n_opt=24, n_fixed=3, n_points=700), dict(seed=14, n_opt=3, n_fixed=2, n_points=300),
                                 dict(seed=15, n_opt=27, n_fixed=2, n_points=700), dict(seed=16, n_opt=29, n_fixed=2, n_points=700),
                                 dict(seed=17, n_opt=30, n_fixed=2, n_points=700)])
def test_local_bundle_adjustment(opt, cfg):
    b = ba_problem(**cfg)
    t0 = time.time()
    its_ref, kp_ref, mp_ref, er_ref = O.local_ba(b["kf_pose"], b["kf_fixed"], b["mp_pos"], b["e_mp"], b["e_kf"], b["e_obs"], b["e_w"], b["K"])
    t_cpu = time.time() - t0
    t0 = time.time()
    stats, kp, mp, er = opt.LocalBundleAdjustment(b["kf_pose"], b["kf_fixed"], b["mp_pos"], b["e_mp"], b["e_kf"], b["e_obs"], b["e_w"], b["K"])
    t_gpu = time.time() - t0
    print(f"LBA {cfg}: edges {len(b['e_mp'])} iters ref/gpu {its_ref}/{stats[0]} cpu {t_cpu*1e3:.1f} ms gpu wall {t_gpu*1e3:.1f} ms dev {opt.stage_ms()[5]:.2f} ms")
    assert stats[0] == its_ref, "number of LM iterations"
    for k in range(len(kp)):
        _pose_close(kp[k], kp_ref[k], f"key-frame {k}")
    assert np.array_equal(kp[b["kf_fixed"] == 1], b["kf_pose"][b["kf_fixed"] == 1]), "fixed key-frames must not move"
    scale = np.maximum(np.linalg.norm(mp_ref, axis=1), 1e-2)
    assert (np.linalg.norm(mp - mp_ref, axis=1) <= RTOL * scale).all(), "landmarks"
    assert np.count_nonzero(er != er_ref) == 0, "erase flags"


def test_local_ba_stop_flag_and_no_fixed(opt):
    from rumi_slam_amd import capi
    b = ba_problem(seed=5, n_opt=4, n_fixed=2, n_points=200)
    stop = np.ones(1, np.uint8)
    stats, kp, mp, er = opt.LocalBundleAdjustment(b["kf_pose"], b["kf_fixed"], b["mp_pos"], b["e_mp"], b["e_kf"], b["e_obs"], b["e_w"], b["K"], stop)
    assert stats[3] == 1 and np.array_equal(kp, b["kf_pose"]) and np.array_equal(mp, b["mp_pos"])   # aborted before optimising
    with pytest.raises(capi.RumiError):
        opt.LocalBundleAdjustment(b["kf_pose"], np.zeros_like(b["kf_fixed"]), b["mp_pos"], b["e_mp"], b["e_kf"], b["e_obs"], b["e_w"], b["K"])


@pytest.mark.parametrize("cfg", [dict(seed=5, n_opt=15, n_fixed=10, n_points=2000, outlier_frac=0.08), dict(seed=6, n_opt=8, n_fixed=0, n_points=600),
                                 dict(seed=7, n_opt=30, n_fixed=12, n_points=1500, outlier_frac=0.2)])
def test_merge_window_bundle_adjustment(opt, cfg):
    """Optimizer::LocalBundleAdjustment(pMainKF, vpAdjustKF, vpFixedKF, pbStopFlag): optimize(5) with Huber, outlier edges to level 1,
    kernels off, optimize(10).  Same iteration counts in both passes, same erase flags, poses / landmarks at 1e-4."""
    b = ba_problem(**cfg)
    a = (b["kf_pose"], b["kf_fixed"], b["mp_pos"], b["e_mp"], b["e_kf"], b["e_obs"], b["e_w"], b["K"])
    its2, kp_ref, mp_ref, er_ref = O.merge_ba(*a)
    stats, kp, mp, er = opt.MergeBundleAdjustment(*a)
    assert (stats[0], stats[3]) == (its2[0], its2[1]), f"LM iterations of the two passes: {stats} vs {its2}"
    assert its2[0] > 0 and its2[1] > 0
    for k in range(len(kp)):
        _pose_close(kp[k], kp_ref[k], f"key-frame {k}")
    scale = np.maximum(np.linalg.norm(mp_ref, axis=1), 1e-2)
    assert (np.linalg.norm(mp - mp_ref, axis=1) <= RTOL * scale).all(), "landmarks"
    assert np.count_nonzero(er != er_ref) == 0, "erase flags"
    if cfg.get("outlier_frac"):
        assert er_ref.sum() > 0
    # differs from the single-pass local BA on the same graph (otherwise the second pass is untested)
    if cfg["n_fixed"] > 0:
        _, kp1, _, _ = opt.LocalBundleAdjustment(*a)
        assert np.abs(kp1 - kp).max() > 0


def test_merge_ba_stop_flag(opt):
    b = ba_problem(seed=8, n_opt=4, n_fixed=2, n_points=200)
    a = (b["kf_pose"], b["kf_fixed"], b["mp_pos"], b["e_mp"], b["e_kf"], b["e_obs"], b["e_w"], b["K"])
    stop = np.ones(1, np.uint8)
    stats, kp, mp, er = opt.MergeBundleAdjustment(*a, stop_flag=stop)
    assert stats[0] == 0 and np.array_equal(kp, b["kf_pose"]) and np.array_equal(mp, b["mp_pos"])


def test_local_ba_degenerate_graphs(opt):
    """All key-frames fixed (structure-only adjustment: the reduced pose system is empty), a landmark with no edge, no edge at all."""
    b = ba_problem(seed=11, n_opt=0, n_fixed=6, n_points=400)
    a = [b["kf_pose"], b["kf_fixed"], b["mp_pos"], b["e_mp"], b["e_kf"], b["e_obs"], b["e_w"], b["K"]]
    its_ref, kp_ref, mp_ref, er_ref = O.local_ba(*a)
    stats, kp, mp, er = opt.LocalBundleAdjustment(*a)
    assert stats[0] == its_ref and np.array_equal(kp, b["kf_pose"])
    scale = np.maximum(np.linalg.norm(mp_ref, axis=1), 1e-2)
    assert (np.linalg.norm(mp - mp_ref, axis=1) <= RTOL * scale).all() and np.array_equal(er, er_ref)
    # one more landmark that nobody observes: it must come back untouched
    b2 = ba_problem(seed=12, n_opt=3, n_fixed=2, n_points=150)
    mp_pos = np.concatenate([b2["mp_pos"], np.array([[1.0, 2.0, 3.0]], np.float32)])
    a2 = [b2["kf_pose"], b2["kf_fixed"], mp_pos, b2["e_mp"], b2["e_kf"], b2["e_obs"], b2["e_w"], b2["K"]]
    its_ref, kp_ref, mp_ref, er_ref = O.local_ba(*a2)
    stats, kp, mp, er = opt.LocalBundleAdjustment(*a2)
    assert stats[0] == its_ref and np.array_equal(mp[-1], mp_pos[-1]) and np.array_equal(er, er_ref)
    for k in range(len(kp)):
        _pose_close(kp[k], kp_ref[k], f"key-frame {k}")
    # no edges
    e0 = np.zeros(0, np.int32)
    stats, kp, mp, er = opt.LocalBundleAdjustment(b2["kf_pose"], b2["kf_fixed"], b2["mp_pos"], e0, e0, np.zeros((0, 2), np.float32), np.zeros(0, np.float32), b2["K"])
    assert np.array_equal(kp, b2["kf_pose"]) and np.array_equal(mp, b2["mp_pos"]) and len(er) == 0


def test_sim3_inlier_scoring(opt):
    """Sim3Solver::ComputeInliersNum: two maps related by a similarity, per key-frame pair the matched points re-projected both
    ways; inlier flags identical, ratios and the returned median identical (the same float divisions)."""
    from scene import quat_from_rotvec, quat_rotate
    rng = np.random.default_rng(4)
    K = K_TUM = np.array([535.4, 539.2, 320.1, 247.6], np.float32)
    n_pairs = 9
    s12 = 1.7
    qS = quat_from_rotvec(np.array([0.05, -0.1, 0.2])); tS = np.array([0.4, -0.2, 1.0])
    _, RS = quat_rotate(qS, np.zeros((1, 3)))
    ps, den, A, B, X1, X2, k1, k2, s1, s2, e1, e2 = [0], [], [], [], [], [], [], [], [], [], [], []

    def compose(qa, ta, sa, qb, tb, sb):            # (sa, Ra, ta) * (sb, Rb, tb)
        _, Ra = quat_rotate(qa, np.zeros((1, 3)))
        x, y, z, w = qa; x2, y2, z2, w2 = qb
        q = np.array([w * x2 + x * w2 + y * z2 - z * y2, w * y2 + y * w2 + z * x2 - x * z2, w * z2 + z * w2 + x * y2 - y * x2, w * w2 - x * x2 - y * y2 - z * z2])
        return q, sa * (Ra @ tb) + ta, sa * sb

    for p in range(n_pairs):
        m = int(rng.integers(0, 120)) if p != 3 else 0
        qc = quat_from_rotvec(rng.normal(size=3) * 0.05); tc = rng.normal(size=3) * 0.1            # camera 1 from world 1
        _, Rc = quat_rotate(qc, np.zeros((1, 3)))
        Pc = np.stack([rng.uniform(-1, 1, m), rng.uniform(-0.8, 0.8, m), rng.uniform(2, 6, m)], 1)   # points in camera 1
        Pw1 = (Pc - tc) @ Rc                                                                         # world 1
        Pw2 = ((Pw1 - tS) @ RS) / s12                                                                # world 2: Pw1 = s R Pw2 + t
        # gSw1w2 = (s12, RS, tS); camera 2 == camera 1 seen from world 2: Sc2w2 = Sc1w1 * Sw1w2 with unit scale kept by rescaling points
        q12, t12, sc = compose(qc, tc, 1.0, qS, tS, s12)                                             # gSc1w2
        A.append(np.concatenate([q12, t12, [sc]]))
        # key-frame 2 pose in world 2 (scale 1): same rotation, translation scaled
        q2, t2 = q12, t12 / s12
        # gSc2w1 = gSc2w2 * gSw1w2^-1
        _, RSi = RS.T, None
        qSi = np.array([-qS[0], -qS[1], -qS[2], qS[3]]); tSi = -(RS.T @ tS) / s12
        qb, tb, sb = compose(q2, t2, 1.0, qSi, tSi, 1.0 / s12)
        B.append(np.concatenate([qb, tb, [sb]]))
        u = K[0] * Pc[:, 0] / Pc[:, 2] + K[2]; v = K[1] * Pc[:, 1] / Pc[:, 2] + K[3]
        noise = rng.normal(0, 1.5, (m, 2)); noise[rng.random(m) < 0.2] += 30
        k1.append(np.stack([u, v], 1) + noise); k2.append(np.stack([u, v], 1) + rng.normal(0, 1.5, (m, 2)))
        X1.append(Pw1); X2.append(Pw2)
        s1.append(np.float32(1.2) ** (2 * rng.integers(0, 8, m))); s2.append(np.float32(1.2) ** (2 * rng.integers(0, 8, m)))
        e1.append((rng.random(m) < 0.1)); e2.append((rng.random(m) < 0.1))
        ps.append(ps[-1] + m); den.append(m + int(rng.integers(0, 3)))
    cat = lambda l, w: np.concatenate(l).reshape(-1, w) if w else np.concatenate(l)
    args = (ps, den, np.stack(A), np.stack(B), K, K, cat(X1, 3), cat(X2, 3), cat(k1, 2), cat(k2, 2), cat(s1, 0), cat(s2, 0), cat(e1, 0), cat(e2, 0))
    med_ref, r_ref, in_ref = O.sim3_inliers(*args)
    med, r, inl = opt.ComputeInliersNum(*args)
    assert 0.3 < in_ref.mean() < 0.99 and len(in_ref) > 300
    assert np.array_equal(inl, in_ref) and np.array_equal(r, r_ref) and med == med_ref


@pytest.mark.parametrize("cfg,its,robust", [(dict(seed=21, n_opt=1, n_fixed=1, n_points=300), 20, True), (dict(seed=22, n_opt=24, n_fixed=1, n_points=1200), 10, False),
                                            (dict(seed=23, n_opt=9, n_fixed=1, n_points=700, outlier_frac=0.1), 5, True)])
def test_global_bundle_adjustment(opt, cfg, its, robust):
    """Optimizer::BundleAdjustment: the 2-key-frame, 20-iteration BA of the monocular map initialisation, and larger windows with
    and without the robust kernel; one fixed key-frame (the map's first)."""
    b = ba_problem(**cfg)
    a = (b["kf_pose"], b["kf_fixed"], b["mp_pos"], b["e_mp"], b["e_kf"], b["e_obs"], b["e_w"], b["K"])
    its_ref, kp_ref, mp_ref = O.bundle_adjustment(*a, its, robust)
    stats, kp, mp = opt.BundleAdjustment(*a, n_iterations=its, robust=robust)
    assert stats[0] == its_ref and its_ref > 1
    for k in range(len(kp)):
        _pose_close(kp[k], kp_ref[k], f"key-frame {k}")
    scale = np.maximum(np.linalg.norm(mp_ref, axis=1), 1e-2)
    assert (np.linalg.norm(mp - mp_ref, axis=1) <= RTOL * scale).all(), "landmarks"


def _sim3_close(a, b, what):
    """Sim3 as (qx qy qz qw tx ty tz s): 1e-4 relative on the rotation as a whole, the translation as a whole and the scale."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert np.linalg.norm(a[:4] - b[:4]) <= 1e-4, f"{what}: rotation {a[:4]} vs {b[:4]}"
    assert np.linalg.norm(a[4:7] - b[4:7]) <= 1e-4 * max(np.linalg.norm(b[4:7]), 1.0), f"{what}: translation {a[4:7]} vs {b[4:7]}"
    assert abs(a[7] - b[7]) <= 1e-4 * abs(b[7]), f"{what}: scale {a[7]} vs {b[7]}"


@pytest.mark.parametrize("cfg,fix", [(dict(seed=1), False), (dict(seed=2, n=900, outlier_frac=0.2), False), (dict(seed=3, n=120, scale=1.0), True),
                                     (dict(seed=4, n=60, outlier_frac=0.0, init_err=0.2), False)])
def test_optimize_sim3_pair(opt, cfg, fix):
    """Optimizer::OptimizeSim3 (loop / merge candidate verification): Huber first pass, numeric Jacobians, outlier removal, second
    pass.  Same inlier set and counts as the oracle, Sim3 within 1e-4."""
    from sim3_scene import sim3_pair_problem
    b = sim3_pair_problem(**cfg)
    a = (b["S0"], b["P1c"], b["P2c"], b["obs1"], b["obs2"], b["w1"], b["w2"], b["K"], b["K"], 10.0, fix, True)
    nin_r, nbad_r, early_r, S_r, st_r = O.optimize_sim3(*a)
    nin, nbad, early, S, st = opt.OptimizeSim3(*a)
    assert nin_r > 0.5 * len(st_r) and not early_r
    assert (nin, nbad, early) == (nin_r, nbad_r, early_r) and np.array_equal(st, st_r)
    _sim3_close(S, S_r, "S12")
    if fix:
        assert S[7] == b["S0"][7]


@pytest.mark.parametrize("seed,fix", [(11, True), (12, False)])
def test_optimize_cloud_sim3(opt, seed, fix):
    """Optimizer::OptimizeCloudSim3 (rumination sub-map alignment): world edges over several key-frame pairs, no robust kernel,
    absent edges for isEdge points, one nearly empty pair."""
    from sim3_scene import sim3_cloud_problem
    c = sim3_cloud_problem(seed=seed)
    a = (c["S0"], c["P1c"], c["P2c"], c["obs1"], c["obs2"], c["w1"], c["w2"], c["K"], c["K"], 10.0, fix, False, c["pair_of"], c["S_c1w"], c["S_c2w"],
         c["skip12"], c["skip21"])
    nin_r, nbad_r, early_r, S_r, st_r = O.optimize_sim3(*a)
    nin, nbad, early, S, st = opt.OptimizeSim3(*a)
    assert nin_r > 0.4 * len(st_r) and (st_r == 3).any() and not early_r
    assert (nin, nbad, early) == (nin_r, nbad_r, early_r) and np.array_equal(st, st_r)
    _sim3_close(S, S_r, "gSw1w2")


def test_optimize_sim3_early_returns(opt):
    """Fewer than 10 surviving correspondences: returns 0 after the first optimize (Optimizer.cc:2135-2136, :2437-2438); no
    correspondences at all: the estimate comes back unchanged."""
    from sim3_scene import sim3_pair_problem
    b = sim3_pair_problem(seed=7, n=9)
    a = (b["S0"], b["P1c"], b["P2c"], b["obs1"], b["obs2"], b["w1"], b["w2"], b["K"], b["K"], 10.0, False, True)
    r_ref, r = O.optimize_sim3(*a), opt.OptimizeSim3(*a)
    assert r_ref[2] and r_ref[0] == 0 and r[:3] == r_ref[:3] and np.array_equal(r[4], r_ref[4])
    _sim3_close(r[3], r_ref[3], "after the first pass")
    # all observations of key-frame 1 are wrong: everything is removed after the first pass
    b = sim3_pair_problem(seed=8, n=200, outlier_frac=1.0)
    a = (b["S0"], b["P1c"], b["P2c"], b["obs1"], b["obs2"], b["w1"], b["w2"], b["K"], b["K"], 10.0, False, True)
    r_ref, r = O.optimize_sim3(*a), opt.OptimizeSim3(*a)
    assert r_ref[2] and r[:3] == r_ref[:3] and np.array_equal(r[4], r_ref[4])
    e = np.zeros((0, 3), np.float32); e2 = np.zeros((0, 2), np.float32); e1 = np.zeros(0, np.float32)
    nin, nbad, early, S, st = opt.OptimizeSim3(b["S0"], e, e, e2, e2, e1, e1, b["K"], b["K"])
    assert (nin, nbad, early) == (0, 0, True) and np.array_equal(S, b["S0"]) and len(st) == 0


@pytest.fixture(scope="module")
def opt_big():
    from rumi_slam_amd.optimizer import Optimizer
    return Optimizer(max_kf=192, max_mp=8192, max_edges=1 << 18)


@pytest.mark.parametrize("cfg,its,robust", [(dict(seed=31, n_opt=60, n_fixed=2, n_points=1500), 10, False),
                                            (dict(seed=32, n_opt=130, n_fixed=1, n_points=2500, outlier_frac=0.08), 6, True)])
def test_large_window_bundle_adjustment(opt_big, cfg, its, robust):
    """Optimizer::BundleAdjustment over more than 42 optimised key-frames (global BA after a loop closure / merge): block-sparse
    Schur accumulation and the multi-workgroup blocked Cholesky instead of the dense-panel path; same iteration count, poses and
    landmarks within 1e-4 of the oracle's dense solve."""
    b = ba_problem(**cfg)
    a = (b["kf_pose"], b["kf_fixed"], b["mp_pos"], b["e_mp"], b["e_kf"], b["e_obs"], b["e_w"], b["K"])
    its_ref, kp_ref, mp_ref = O.bundle_adjustment(*a, its, robust)
    stats, kp, mp = opt_big.BundleAdjustment(*a, n_iterations=its, robust=robust)
    assert stats[2] == cfg["n_opt"] and 6 * cfg["n_opt"] > 255
    assert stats[0] == its_ref, f"LM iterations {stats[0]} vs {its_ref}"
    for k in range(len(kp)):
        _pose_close(kp[k], kp_ref[k], f"key-frame {k}")
    rel = np.linalg.norm(mp - mp_ref, axis=1) / np.maximum(np.linalg.norm(mp_ref, axis=1), 1e-3)
    assert rel.max() <= RTOL, f"landmark rel diff {rel.max()}"
    assert np.abs(kp[~b["kf_fixed"].astype(bool)] - b["kf_pose"][~b["kf_fixed"].astype(bool)]).max() > 1e-3      # the poses did move


def test_large_window_local_ba(opt_big):
    """LocalBundleAdjustment with 50 optimised key-frames (a dense local map) goes down the same large-window path."""
    b = ba_problem(seed=33, n_opt=50, n_fixed=4, n_points=1200)
    a = (b["kf_pose"], b["kf_fixed"], b["mp_pos"], b["e_mp"], b["e_kf"], b["e_obs"], b["e_w"], b["K"])
    its_ref, kp_ref, mp_ref, er_ref = O.local_ba(*a)
    stats, kp, mp, er = opt_big.LocalBundleAdjustment(*a)
    assert stats[0] == its_ref and np.array_equal(er, er_ref)
    for k in range(len(kp)):
        _pose_close(kp[k], kp_ref[k], f"key-frame {k}")
    rel = np.linalg.norm(mp - mp_ref, axis=1) / np.maximum(np.linalg.norm(mp_ref, axis=1), 1e-3)
    assert rel.max() <= RTOL


def test_error_conventions_of_the_new_entry_points(opt):
    """Status codes instead of exceptions across the ABI: a key-frame-pair index outside the given pairs is RUMI_E_INVALID, a window larger
    than the handle's arenas is RUMI_E_CAPACITY, and neither leaves the handle unusable."""
    from rumi_slam_amd import capi
    from rumi_slam_amd.optimizer import Optimizer
    from sim3_scene import sim3_cloud_problem, sim3_pair_problem
    c = sim3_cloud_problem(seed=5, n_pairs=3, per_pair=40)
    bad = c["pair_of"].copy(); bad[7] = 3
    with pytest.raises(capi.RumiError) as e:
        opt.OptimizeSim3(c["S0"], c["P1c"], c["P2c"], c["obs1"], c["obs2"], c["w1"], c["w2"], c["K"], c["K"], 10.0, True, False, bad, c["S_c1w"], c["S_c2w"])
    assert e.value.code == capi.RUMI_E_INVALID
    small = Optimizer(max_kf=8, max_mp=256, max_edges=4096)
    b = ba_problem(seed=9, n_opt=10, n_fixed=1, n_points=100)
    with pytest.raises(capi.RumiError) as e:
        small.BundleAdjustment(b["kf_pose"], b["kf_fixed"], b["mp_pos"], b["e_mp"], b["e_kf"], b["e_obs"], b["e_w"], b["K"], n_iterations=3, robust=True)
    assert e.value.code == capi.RUMI_E_CAPACITY
    s = sim3_pair_problem(seed=6, n=60)                      # the handles still work
    r = opt.OptimizeSim3(s["S0"], s["P1c"], s["P2c"], s["obs1"], s["obs2"], s["w1"], s["w2"], s["K"], s["K"])
    assert r[0] > 30
    b2 = ba_problem(seed=9, n_opt=4, n_fixed=1, n_points=100)
    stats, _, _ = small.BundleAdjustment(b2["kf_pose"], b2["kf_fixed"], b2["mp_pos"], b2["e_mp"], b2["e_kf"], b2["e_obs"], b2["e_w"], b2["K"], n_iterations=3, robust=True)
    assert stats[2] == 4


def test_local_ba_batch_equals_single_calls(opt):
    """rumi_local_ba_batch: R independent windows over worker threads with their own child handles and streams; every window's result must be
    the single-call result BIT FOR BIT (round 4: the window is a batch dimension of the kernels and every sum has a fixed order), whatever
    the number of workers asked for."""
    cfgs = [dict(seed=20 + i, n_opt=6 + 3 * (i % 4), n_fixed=2, n_points=300 + 100 * (i % 3), outlier_frac=0.05 * (i % 2)) for i in range(10)]
    probs = [ba_problem(**c) for c in cfgs]
    wins = [(b["kf_pose"], b["kf_fixed"], b["mp_pos"], b["e_mp"], b["e_kf"], b["e_obs"], b["e_w"], b["K"]) for b in probs]
    single = [opt.LocalBundleAdjustment(*w) for w in wins]
    for workers in (1, 3, 4):
        got = opt.LocalBundleAdjustmentBatch(wins, workers)
        for i, (s, g) in enumerate(zip(single, got)):
            assert np.array_equal(s[0], g[0]), f"window {i}, {workers} workers: stats {g[0]} vs {s[0]}"
            assert np.array_equal(s[3], g[3]), f"window {i}, {workers} workers: erase flags"
            assert np.array_equal(s[1], g[1]) and np.array_equal(s[2], g[2]), f"window {i}, {workers} workers: values differ from the single call"


def test_local_ba_batch_of_more_windows_than_one_launch_takes(opt):
    """35 windows in one call: more than the 32 a launch group's table holds (the entry cuts the batch), the first cut runs as three launch groups;
    every window bit for bit the single call, in the caller's order."""
    cfgs = [dict(seed=500 + i, n_opt=3 + (i % 5), n_fixed=2, n_points=150 + 20 * (i % 4), outlier_frac=0.04 * (i % 2)) for i in range(35)]
    probs = [ba_problem(**c) for c in cfgs]
    wins = [(b["kf_pose"], b["kf_fixed"], b["mp_pos"], b["e_mp"], b["e_kf"], b["e_obs"], b["e_w"], b["K"]) for b in probs]
    single = [opt.LocalBundleAdjustment(*w) for w in wins]
    got = opt.LocalBundleAdjustmentBatch(wins, 1)
    assert len(got) == len(wins)
    for i, (s_, g) in enumerate(zip(single, got)):
        assert np.array_equal(s_[0], g[0]) and np.array_equal(s_[3], g[3]), f"window {i}: stats / erase flags"
        assert np.array_equal(s_[1], g[1]) and np.array_equal(s_[2], g[2]), f"window {i}: values differ from the single call"


def test_local_ba_batch_windows_of_very_different_size(opt):
    """Windows of 6 to 29 optimised key-frames in ONE rumi_local_ba_batch call over several workers: the solve kernels of the larger windows need
    74-135 KB of dynamic LDS, which is an opt-in kept per FUNCTION, i.e. shared by all worker threads.  A worker with a small window must not
    lower the limit under a worker about to launch a large one (the limit only grows: rumi_common.h raise_lds_limit); repeated so that the
    workers meet in different orders."""
    sizes = [29, 6, 22, 8, 16, 27, 7, 19, 6, 24, 9, 17]
    cfgs = [dict(seed=300 + i, n_opt=k, n_fixed=2, n_points=400 + 50 * (i % 4), outlier_frac=0.03) for i, k in enumerate(sizes)]
    probs = [ba_problem(**c) for c in cfgs]
    wins = [(b["kf_pose"], b["kf_fixed"], b["mp_pos"], b["e_mp"], b["e_kf"], b["e_obs"], b["e_w"], b["K"]) for b in probs]
    single = [opt.LocalBundleAdjustment(*w) for w in wins]
    for rep in range(3):
        order = list(range(len(wins))) if rep == 0 else list(np.random.default_rng(rep).permutation(len(wins)))
        got = opt.LocalBundleAdjustmentBatch([wins[i] for i in order], 4)
        for j, i in enumerate(order):
            s, g = single[i], got[j]
            assert np.array_equal(s[0], g[0]), f"window {i} ({sizes[i]} key-frames), round {rep}: stats {g[0]} vs {s[0]}"
            assert np.array_equal(s[3], g[3]), f"window {i}, round {rep}: erase flags"
            assert np.array_equal(s[1], g[1]) and np.array_equal(s[2], g[2]), f"window {i}, round {rep}: values differ from the single call"


@pytest.mark.parametrize("cfg", [dict(seed=0, n_opt=20, n_fixed=5, n_points=3000), dict(seed=16, n_opt=29, n_fixed=2, n_points=700),
                                 dict(seed=3, n_opt=12, n_fixed=1, n_points=1500, outlier_frac=0.15)])
def test_local_ba_is_bit_reproducible(opt, cfg):
    """No floating-point atomics on anything the Levenberg-Marquardt decisions read (chi2, H_ll / b_l, H_pp / b_p, the Schur complement, the gain
    denominator are summed in a fixed order): BASELINE configs[3] and the largest window of the tile solver, three runs each and once more inside a
    batch of other windows -- outputs, erase flags, LM iteration AND trial counts identical bit for bit."""
    b = ba_problem(**cfg)
    a = (b["kf_pose"], b["kf_fixed"], b["mp_pos"], b["e_mp"], b["e_kf"], b["e_obs"], b["e_w"], b["K"])
    runs = [opt.LocalBundleAdjustment(*a) for _ in range(3)]
    other = [ba_problem(seed=40 + i, n_opt=5 + 4 * i, n_fixed=2, n_points=300 + 150 * i) for i in range(3)]
    wins = [(o["kf_pose"], o["kf_fixed"], o["mp_pos"], o["e_mp"], o["e_kf"], o["e_obs"], o["e_w"], o["K"]) for o in other]
    runs.append(opt.LocalBundleAdjustmentBatch(wins[:2] + [a] + wins[2:], 1)[2])
    for r in runs[1:]:
        assert np.array_equal(r[0], runs[0][0]), f"stats {r[0]} vs {runs[0][0]}"
        assert r[1].tobytes() == runs[0][1].tobytes() and r[2].tobytes() == runs[0][2].tobytes() and np.array_equal(r[3], runs[0][3])
    assert runs[0][0][1] >= runs[0][0][0] > 0


def test_merge_ba_is_bit_reproducible(opt):
    b = ba_problem(seed=5, n_opt=15, n_fixed=10, n_points=2000, outlier_frac=0.08)
    a = (b["kf_pose"], b["kf_fixed"], b["mp_pos"], b["e_mp"], b["e_kf"], b["e_obs"], b["e_w"], b["K"])
    runs = [opt.MergeBundleAdjustment(*a) for _ in range(3)]
    for r in runs[1:]:
        assert np.array_equal(r[0], runs[0][0]) and r[1].tobytes() == runs[0][1].tobytes() and r[2].tobytes() == runs[0][2].tobytes() and np.array_equal(r[3], runs[0][3])
