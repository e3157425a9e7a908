"""Synthetic optimisation problems: key-frames on a smooth arc looking at a box of points,
pixel noise by octave, gross outliers, perturbed initial estimates.  TEST INFRASTRUCTURE (numpy only)."""
import numpy as np

from scene import K_TUM3, quat_from_rotvec, quat_rotate


def _project(q, t, X, K):
    Xc, _ = quat_rotate(q, X)
    Xc = Xc + t
    u = K[0] * Xc[:, 0] / Xc[:, 2] + K[2]
    v = K[1] * Xc[:, 1] / Xc[:, 2] + K[3]
    return u, v, Xc[:, 2]


def ba_problem(seed=0, n_opt=20, n_fixed=5, n_points=3000, outlier_frac=0.05, w=640, h=480, pose_noise=(np.deg2rad(1.0), 0.02),
               point_noise=0.03):
    rng = np.random.default_rng(seed)
    K = K_TUM3.astype(np.float64)
    nkf = n_opt + n_fixed
    # camera centres on an arc of radius 3 m, 2 degree steps, looking along +z with a slow yaw
    poses_q, poses_t = [], []
    for k in range(nkf):
        a = np.deg2rad(2.0) * (k - nkf / 2)
        q_wc = quat_from_rotvec(np.array([0.0, a, 0.0]))             # camera-to-world rotation
        c = np.array([3.0 * np.sin(a), 0.02 * k, 3.0 * (1 - np.cos(a))])
        q_cw = q_wc * np.array([-1, -1, -1, 1.0])
        t_cw = -quat_rotate(q_cw, c[None])[0][0]
        poses_q.append(q_cw); poses_t.append(t_cw)
    X = np.stack([rng.uniform(-3, 3, n_points), rng.uniform(-2, 2, n_points), rng.uniform(2, 8, n_points)], 1)
    e_mp, e_kf, e_obs, e_w = [], [], [], []
    for p in range(n_points):                                       # edges grouped by point, as LocalBundleAdjustment builds them
        for k in range(nkf):
            u, v, z = _project(poses_q[k], poses_t[k], X[p:p + 1], K)
            if z[0] > 0.1 and 0 <= u[0] < w and 0 <= v[0] < h:
                octave = int(rng.integers(0, 8))
                sig = 1.2 ** octave
                du, dv = rng.normal(size=2) * sig
                if rng.random() < outlier_frac:
                    du += rng.uniform(10, 50) * rng.choice([-1, 1]); dv += rng.uniform(10, 50) * rng.choice([-1, 1])
                e_mp.append(p); e_kf.append(k); e_obs.append((u[0] + du, v[0] + dv)); e_w.append(1.0 / (np.float32(1.2) ** np.float32(2 * octave)))
    # perturbed initial estimates (fixed key-frames keep the truth)
    kf_pose = np.zeros((nkf, 7), np.float32)
    fixed = np.zeros(nkf, np.uint8); fixed[:n_fixed] = 1
    for k in range(nkf):
        q, t = poses_q[k], poses_t[k]
        if not fixed[k]:
            dq = quat_from_rotvec(rng.normal(size=3) * pose_noise[0] / np.sqrt(3))
            x1, y1, z1, w1 = dq; x2, y2, z2, w2 = q
            q = np.array([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2,
                          w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2])
            t = t + rng.normal(size=3) * pose_noise[1] / np.sqrt(3)
        kf_pose[k] = np.concatenate([q, t])
    mp = (X + rng.normal(size=X.shape) * point_noise / np.sqrt(3)).astype(np.float32)
    return dict(kf_pose=kf_pose, kf_fixed=fixed, mp_pos=mp, e_mp=np.array(e_mp, np.int32), e_kf=np.array(e_kf, np.int32),
                e_obs=np.array(e_obs, np.float32), e_w=np.array(e_w, np.float32), K=K_TUM3.copy(),
                truth_q=np.array(poses_q), truth_t=np.array(poses_t), truth_X=X)


def pose_problem(seed=0, n=300, outlier_frac=0.1, w=640, h=480):
    rng = np.random.default_rng(seed)
    K = K_TUM3.astype(np.float64)
    q = quat_from_rotvec(rng.normal(size=3) * 0.2); t = rng.normal(size=3) * 0.3
    Xc = np.stack([rng.uniform(-2.5, 2.5, 4 * n), rng.uniform(-1.8, 1.8, 4 * n), rng.uniform(1.5, 8, 4 * n)], 1)
    u = K[0] * Xc[:, 0] / Xc[:, 2] + K[2]; v = K[1] * Xc[:, 1] / Xc[:, 2] + K[3]
    ok = (u >= 0) & (u < w) & (v >= 0) & (v < h)
    Xc, u, v = Xc[ok][:n], u[ok][:n], v[ok][:n]
    n = len(Xc)
    _, R = quat_rotate(q, np.zeros((1, 3)))
    Xw = (Xc - t) @ R
    octave = rng.integers(0, 8, n)
    sig = 1.2 ** octave
    obs = np.stack([u + rng.normal(size=n) * sig, v + rng.normal(size=n) * sig], 1)
    bad = rng.random(n) < outlier_frac
    obs[bad] += rng.uniform(10, 50, (bad.sum(), 2)) * rng.choice([-1, 1], (bad.sum(), 2))
    inv_sigma2 = (1.0 / (np.float32(1.2) ** (2 * octave).astype(np.float32))).astype(np.float32)
    dq = quat_from_rotvec(rng.normal(size=3) * 0.02)
    x1, y1, z1, w1 = dq; x2, y2, z2, w2 = q
    q0 = np.array([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2, w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2,
                   w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2])
    T0 = np.concatenate([q0, t + rng.normal(size=3) * 0.05]).astype(np.float32)
    return dict(Xw=Xw.astype(np.float32), obs=obs.astype(np.float32), inv_sigma2=inv_sigma2, K=K_TUM3.copy(), T0=T0,
                truth=np.concatenate([q, t]), bad=bad)
