"""Synthetic problems for Optimizer::OptimizeSim3 / OptimizeCloudSim3: two maps related by a similarity, matched map points seen
from one key-frame of each map (single pair) or from several key-frame pairs (cloud variant), pixel noise by octave, gross outliers."""
import numpy as np

from scene import quat_from_rotvec, quat_rotate

K_TUM = np.array([535.4, 539.2, 320.1, 247.6], np.float32)


def _R(q):
    return quat_rotate(q, np.zeros((1, 3)))[1]


def _qmul(a, b):
    x, y, z, w = a; x2, y2, z2, w2 = b
    return np.array([w * x2 + x * w2 + y * z2 - z * y2, w * y2 + y * w2 + z * x2 - x * z2, w * z2 + z * w2 + x * y2 - y * x2, w * w2 - x * x2 - y * y2 - z * z2])


def _project(P):
    return np.stack([K_TUM[0] * P[:, 0] / P[:, 2] + K_TUM[2], K_TUM[1] * P[:, 1] / P[:, 2] + K_TUM[3]], 1)


def sim3_pair_problem(seed=0, n=400, scale=1.3, outlier_frac=0.1, init_err=1.0, noise=1.0):
    """OptimizeSim3: the vertex S12 maps camera-2 coordinates into camera-1 coordinates."""
    rng = np.random.default_rng(seed)
    q = quat_from_rotvec(np.array([0.04, -0.12, 0.07])); t = np.array([0.3, -0.1, 0.25]); R = _R(q)
    P2 = np.stack([rng.uniform(-1.5, 1.5, n), rng.uniform(-1.0, 1.0, n), rng.uniform(2.5, 7, n)], 1)
    P1 = scale * (P2 @ R.T) + t
    oct1, oct2 = rng.integers(0, 8, n), rng.integers(0, 8, n)
    o1 = _project(P1) + rng.normal(0, 1, (n, 2)) * (noise * 1.2 ** oct1)[:, None]
    o2 = _project(P2) + rng.normal(0, 1, (n, 2)) * (noise * 1.2 ** oct2)[:, None]
    bad = rng.random(n) < outlier_frac
    o1[bad] += rng.uniform(15, 60, (int(bad.sum()), 2)) * rng.choice([-1, 1], (int(bad.sum()), 2))
    # the map points themselves are not perfectly consistent either
    P1n = P1 + rng.normal(0, 0.01, (n, 3)); P2n = P2 + rng.normal(0, 0.01, (n, 3))
    q0 = _qmul(quat_from_rotvec(rng.normal(size=3) * 0.01 * init_err), q)
    S0 = np.concatenate([q0, t + rng.normal(size=3) * 0.03 * init_err, [scale * (1 + 0.03 * init_err)]])
    w1 = (np.float32(1.2) ** (-2 * oct1)).astype(np.float32); w2 = (np.float32(1.2) ** (-2 * oct2)).astype(np.float32)
    return dict(S0=S0, S_true=np.concatenate([q, t, [scale]]), P1c=P1n.astype(np.float32), P2c=P2n.astype(np.float32), obs1=o1.astype(np.float32),
                obs2=o2.astype(np.float32), w1=w1, w2=w2, K=K_TUM, bad=bad)


def sim3_cloud_problem(seed=0, n_pairs=6, per_pair=150, scale=1.0, outlier_frac=0.1, init_err=1.0, edge_frac=0.05):
    """OptimizeCloudSim3: the vertex gSw1w2 maps world-2 coordinates into world-1 coordinates; per key-frame pair gSc1w, gSc2w."""
    rng = np.random.default_rng(seed)
    qS = quat_from_rotvec(np.array([0.03, 0.15, -0.05])); tS = np.array([0.5, 0.1, -0.3]); RS = _R(qS)
    A, B, pair_of, P1c, P2c, o1, o2, w1, w2 = [], [], [], [], [], [], [], [], []
    for p in range(n_pairs):
        m = per_pair + int(rng.integers(-20, 20)) if p != 2 else 3
        q1 = quat_from_rotvec(rng.normal(size=3) * 0.08); t1 = rng.normal(size=3) * 0.3; R1 = _R(q1)          # camera 1 from world 1
        q2 = quat_from_rotvec(rng.normal(size=3) * 0.08); R2 = _R(q2)
        # put camera 2 where it sees the same points: world-2 points = S^-1 (world-1 points)
        Pc1 = np.stack([rng.uniform(-1.5, 1.5, m), rng.uniform(-1, 1, m), rng.uniform(3, 8, m)], 1)
        Pw1 = (Pc1 - t1) @ R1
        Pw2 = ((Pw1 - tS) @ RS) / scale
        c2 = Pw2.mean(0) - R2.T @ np.array([0, 0, 5.0]) if m else np.zeros(3)
        t2 = -R2 @ c2
        Pc2 = Pw2 @ R2.T + t2
        A.append(np.concatenate([q1, t1, [1.0]])); B.append(np.concatenate([q2, t2, [1.0]]))
        oc1, oc2 = rng.integers(0, 8, m), rng.integers(0, 8, m)
        a = _project(Pc1) + rng.normal(0, 1, (m, 2)) * (1.2 ** oc1)[:, None]
        b = _project(Pc2) + rng.normal(0, 1, (m, 2)) * (1.2 ** oc2)[:, None]
        bad = rng.random(m) < outlier_frac
        b[bad] += rng.uniform(15, 60, (int(bad.sum()), 2)) * rng.choice([-1, 1], (int(bad.sum()), 2))
        pair_of += [p] * m
        P1c.append(Pc1 + rng.normal(0, 0.01, (m, 3))); P2c.append(Pc2 + rng.normal(0, 0.01, (m, 3))); o1.append(a); o2.append(b)
        w1.append(np.float32(1.2) ** (-2 * oc1)); w2.append(np.float32(1.2) ** (-2 * oc2))
    n = len(pair_of)
    q0 = _qmul(quat_from_rotvec(rng.normal(size=3) * 0.01 * init_err), qS)
    S0 = np.concatenate([q0, tS + rng.normal(size=3) * 0.05 * init_err, [scale]])
    cat = lambda l, w: np.concatenate(l).reshape(-1, w).astype(np.float32) if w else np.concatenate(l).astype(np.float32)
    return dict(S0=S0, S_true=np.concatenate([qS, tS, [scale]]), pair_of=np.array(pair_of, np.int32), S_c1w=np.stack(A), S_c2w=np.stack(B), P1c=cat(P1c, 3),
                P2c=cat(P2c, 3), obs1=cat(o1, 2), obs2=cat(o2, 2), w1=cat(w1, 0), w2=cat(w2, 0), K=K_TUM,
                skip12=(rng.random(n) < edge_frac).astype(np.uint8), skip21=(rng.random(n) < edge_frac).astype(np.uint8))


def sim3_ransac_problem(seed=0, n_pairs=5, per_pair=120, n_solver=90, scale=1.25, outlier_frac=0.3):
    """Sim3Solver::iterate of the sub-map merge: the solver's key-frame pair with matched map points in the two camera frames (a share of
    them wrong matches), and the score set of ComputeInliersNum: every key-frame pair's matched key-points with the points' world positions."""
    rng = np.random.default_rng(seed)
    qS = quat_from_rotvec(np.array([0.05, -0.2, 0.08])); tS = np.array([0.4, -0.2, 0.3]); RS = _R(qS)

    def pair(m):
        q1 = quat_from_rotvec(rng.normal(size=3) * 0.08); t1 = rng.normal(size=3) * 0.3; R1 = _R(q1)
        q2 = quat_from_rotvec(rng.normal(size=3) * 0.08); R2 = _R(q2)
        Pc1 = np.stack([rng.uniform(-1.5, 1.5, m), rng.uniform(-1, 1, m), rng.uniform(3, 8, m)], 1)
        Pw1 = (Pc1 - t1) @ R1
        Pw2 = ((Pw1 - tS) @ RS) / scale
        c2 = Pw2.mean(0) - R2.T @ np.array([0, 0, 5.0 / scale])
        t2 = -R2 @ c2
        return np.concatenate([q1, t1, [1.0]]), np.concatenate([q2, t2, [1.0]]), Pc1, Pw1, Pw2, Pw2 @ R2.T + t2

    S1, S2, Pc1, _, _, Pc2 = pair(n_solver)
    X1 = Pc1 + rng.normal(0, 0.004, Pc1.shape); X2 = Pc2 + rng.normal(0, 0.004, Pc2.shape)
    bad = rng.random(n_solver) < outlier_frac
    X2[bad] = np.stack([rng.uniform(-1.5, 1.5, int(bad.sum())), rng.uniform(-1, 1, int(bad.sum())), rng.uniform(2, 6, int(bad.sum()))], 1)
    sg = lambda m: (np.float32(1.2) ** (2 * rng.integers(0, 8, m))).astype(np.float32)
    A, B, ps, pd, W1, W2, k1, k2 = [S1], [S2], [0], [], [], [], [], []
    for p in range(n_pairs):
        m = per_pair + int(rng.integers(-15, 15))
        a, b, c1, w1, w2, c2 = pair(m)
        if p:
            A.append(a); B.append(b)
        else:                                   # pair 0 is seen from the solver's own key-frames
            a, b = S1, S2
            R1, R2 = _R(S1[:4]), _R(S2[:4])
            c1, c2 = w1 @ R1.T + S1[4:7], w2 @ R2.T + S2[4:7]
        o1 = _project(c1) + rng.normal(0, 0.7, (m, 2)); o2 = _project(c2) + rng.normal(0, 0.7, (m, 2))
        wrong = rng.random(m) < 0.2
        o2[wrong] += rng.uniform(25, 80, (int(wrong.sum()), 2)) * rng.choice([-1, 1], (int(wrong.sum()), 2))
        W1.append(w1); W2.append(w2); k1.append(o1); k2.append(o2)
        ps.append(ps[-1] + m); pd.append(m + int(rng.integers(0, 10)))
    total = ps[-1]
    score = dict(pair_start=np.array(ps, np.int32), pair_denominator=np.array(pd, np.int32), S_c1w1=np.stack(A), S_c2w2=np.stack(B), S_kf1w=S1, S_kf2w=S2,
                 K4_1=K_TUM, K4_2=K_TUM, X1=np.concatenate(W1).astype(np.float32), X2=np.concatenate(W2).astype(np.float32),
                 kp1=np.concatenate(k1).astype(np.float32), kp2=np.concatenate(k2).astype(np.float32), sigma2_1=sg(total), sigma2_2=sg(total),
                 edge1=(rng.random(total) < 0.03).astype(np.uint8), edge2=(rng.random(total) < 0.03).astype(np.uint8))
    # gSc1c2 that the solver should find: camera-2 coordinates -> camera-1 coordinates
    R1, R2 = _R(S1[:4]), _R(S2[:4])
    R12 = R1 @ RS @ R2.T
    t12 = R1 @ (tS - scale * RS @ R2.T @ S2[4:7]) + S1[4:7]
    return dict(X1=X1.astype(np.float32), X2=X2.astype(np.float32), sigma2_1=sg(n_solver), sigma2_2=sg(n_solver), K=K_TUM, bad=bad, score=score,
                R12=R12, t12=t12, s12=scale)
