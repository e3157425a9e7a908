"""CPU tests of the optimiser oracle (test infrastructure): the restated g2o Levenberg-Marquardt must behave like an
optimiser — converge to the truth on clean data, reject gross outliers, respect the reference's early exits."""
import numpy as np

import oracle_lib as O
from ba_scene import ba_problem, pose_problem
from scene import quat_rotate


def _reproj(q, t, X, K):
    Xc, _ = quat_rotate(q.astype(np.float64), X.astype(np.float64))
    Xc = Xc + t
    return np.stack([K[0] * Xc[:, 0] / Xc[:, 2] + K[2], K[1] * Xc[:, 1] / Xc[:, 2] + K[3]], 1)


def test_pose_optimization_recovers_clean_pose():
    p = pose_problem(1, 400, outlier_frac=0.0)
    obs = _reproj(p["truth"][:4], p["truth"][4:], p["Xw"], p["K"]).astype(np.float32)      # noise-free observations
    ng, T, out = O.pose_optimization(p["Xw"], obs, p["inv_sigma2"], p["K"], p["T0"])
    assert ng == len(obs) and not out.any()
    assert np.abs(T - p["truth"]).max() < 2e-4


def test_pose_optimization_flags_gross_outliers():
    p = pose_problem(2, 500, outlier_frac=0.2)
    ng, T, out = O.pose_optimization(p["Xw"], p["obs"], p["inv_sigma2"], p["K"], p["T0"])
    assert out[p["bad"]].mean() > 0.97                      # gross outliers (10-50 px) are rejected
    assert out[~p["bad"]].mean() < 0.12                     # chi2 > 5.991 rejects ~5 % of inliers by construction
    assert ng == len(out) - out.sum()
    assert np.abs(T[4:] - p["truth"][4:]).max() < 0.03


def test_pose_optimization_early_exits():
    p = pose_problem(3, 300)
    ng, T, out = O.pose_optimization(p["Xw"][:2], p["obs"][:2], p["inv_sigma2"][:2], p["K"], p["T0"])
    assert ng == 0 and np.array_equal(T, p["T0"])           # < 3 correspondences: Optimizer.cc:899-900
    ng, T, out = O.pose_optimization(p["Xw"][:8], p["obs"][:8], p["inv_sigma2"][:8], p["K"], p["T0"])
    assert 0 <= ng <= 8                                     # < 10 edges: a single round (:989-990)


def test_local_ba_reduces_reprojection_error_and_keeps_fixed_kfs():
    b = ba_problem(seed=4, n_opt=5, n_fixed=2, n_points=400, outlier_frac=0.05)

    def rms(kf_pose, mp):
        r = []
        for e in range(len(b["e_mp"])):
            k = b["e_kf"][e]
            uv = _reproj(kf_pose[k, :4], kf_pose[k, 4:].astype(np.float64), mp[b["e_mp"][e]][None], b["K"])[0]
            r.append(np.linalg.norm(uv - b["e_obs"][e]) * np.sqrt(b["e_w"][e]))
        return float(np.sqrt(np.median(np.square(r))))

    its, kp, mp, erase = O.local_ba(b["kf_pose"], b["kf_fixed"], b["mp_pos"], b["e_mp"], b["e_kf"], b["e_obs"], b["e_w"], b["K"])
    assert 1 <= its <= 10
    assert rms(kp, mp) < 0.5 * rms(b["kf_pose"], b["mp_pos"])
    assert np.array_equal(kp[b["kf_fixed"] == 1], b["kf_pose"][b["kf_fixed"] == 1])
    assert 0.02 < erase.mean() < 0.2                        # the 5 % gross outliers (+ a few chi2 tail inliers)
    err_before = np.abs(b["kf_pose"][:, 4:] - b["truth_t"]).max()
    err_after = np.abs(kp[:, 4:] - b["truth_t"]).max()
    assert err_after < err_before


def test_local_ba_stop_flag_aborts_before_optimising():
    b = ba_problem(seed=5, n_opt=3, n_fixed=1, n_points=100)
    stop = np.ones(1, np.uint8)
    its, kp, mp, erase = O.local_ba(b["kf_pose"], b["kf_fixed"], b["mp_pos"], b["e_mp"], b["e_kf"], b["e_obs"], b["e_w"], b["K"], stop)
    assert its == -1 and np.array_equal(kp, b["kf_pose"]) and np.array_equal(mp, b["mp_pos"])


def test_optimize_sim3_oracle_recovers_similarity_and_rejects_outliers():
    """OptimizeSim3 restatement (numeric Jacobians through the Sim3 exponential): converges to the true similarity, removes the
    gross outliers after the first pass, keeps the scale when it is fixed."""
    from sim3_scene import sim3_pair_problem
    b = sim3_pair_problem(seed=1, n=500, outlier_frac=0.1)
    a = (b["S0"], b["P1c"], b["P2c"], b["obs1"], b["obs2"], b["w1"], b["w2"], b["K"], b["K"])
    nin, nbad, early, S, st = O.optimize_sim3(*a, 10.0, False, True)
    assert not early and nin + nbad + (st == 2).sum() == 500
    assert (st[b["bad"]] == 1).mean() > 0.95 and (st[~b["bad"]] == 0).mean() > 0.9
    assert np.abs(S[:4] - b["S_true"][:4]).max() < 2e-3 and np.abs(S[4:7] - b["S_true"][4:7]).max() < 2e-2 and abs(S[7] - 1.3) < 1e-2
    nin, nbad, early, S, st = O.optimize_sim3(*a, 10.0, True, True)
    assert S[7] == b["S0"][7]                                 # update[6] = 0: exp(0) * s is exact


def test_optimize_cloud_sim3_oracle_world_edges():
    """OptimizeCloudSim3 restatement: the vertex lives between the two worlds, every key-frame pair wraps it in its own poses."""
    from sim3_scene import sim3_cloud_problem
    c = sim3_cloud_problem(seed=2, outlier_frac=0.03)
    nin, nbad, early, S, st = O.optimize_sim3(c["S0"], c["P1c"], c["P2c"], c["obs1"], c["obs2"], c["w1"], c["w2"], c["K"], c["K"], 10.0, True, False,
                                              c["pair_of"], c["S_c1w"], c["S_c2w"], c["skip12"], c["skip21"])
    assert not early and nin > 0.6 * len(st) and (st == 3).sum() > 0
    assert np.abs(S[:4] - c["S_true"][:4]).max() < 2e-3 and np.abs(S[4:7] - c["S_true"][4:7]).max() < 2e-2 and S[7] == 1.0
