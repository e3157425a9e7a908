"""Independent cross-checks of the oracle with what IS importable in this image (numpy, scipy, torch) — the reference cannot be built here and
holds no fixtures, so these do not pin the OpenCV variants bit for bit, but they bound what the restatement can be getting wrong:
  * resize: the 11-bit fixed-point bilinear is within 1 grey level of float bilinear with cv's pixel-centre mapping;
  * both GaussianBlur variants are within 2 (variant 1: 3) grey levels of the float 7-tap sigma-2 Gaussian with mirror (REFLECT_101) borders,
    and of each other;
  * fastAtan2 is within its documented 0.3 degrees of atan2;
  * PoseOptimization ends in a minimum of the inlier cost that scipy.optimize.least_squares cannot lower (relative 1e-6), i.e. residuals,
    Jacobians and the SE3 update are consistent with an independent solver; LocalBundleAdjustment's 10-iteration LM lowers the Huber cost and
    ends within 2 % of the minimum scipy reaches when started from its end state.
"""
import numpy as np
import pytest
import scipy.ndimage as ndi
import scipy.optimize as sopt

import oracle_lib as O
from ba_scene import ba_problem, pose_problem
from rumi_slam_amd.synth import synth_frame


def _float_bilinear(src, dw, dh):
    sh, sw = src.shape
    fx = (np.arange(dw) + 0.5) * (sw / dw) - 0.5
    fy = (np.arange(dh) + 0.5) * (sh / dh) - 0.5
    x0 = np.floor(fx).astype(int); ax = fx - x0
    y0 = np.floor(fy).astype(int); ay = fy - y0
    xa, xb = np.clip(x0, 0, sw - 1), np.clip(x0 + 1, 0, sw - 1)
    ya, yb = np.clip(y0, 0, sh - 1), np.clip(y0 + 1, 0, sh - 1)
    s = src.astype(np.float64)
    top = s[ya][:, xa] * (1 - ax) + s[ya][:, xb] * ax
    bot = s[yb][:, xa] * (1 - ax) + s[yb][:, xb] * ax
    return top * (1 - ay)[:, None] + bot * ay[:, None]


@pytest.mark.parametrize("size", [(640, 480, 533, 400), (533, 400, 444, 333), (214, 161, 179, 134), (752, 480, 627, 400)])
def test_resize_within_one_level_of_float_bilinear(size):
    sw, sh, dw, dh = size
    img = synth_frame(21, w=sw, h=sh)
    got = O.resize_linear(img, dw, dh).astype(np.float64)
    ref = _float_bilinear(img, dw, dh)
    assert np.max(np.abs(got - ref)) <= 1.0 + 1e-9
    assert np.mean(np.abs(got - ref)) < 0.35                        # (cv's 8-bit path truncates its intermediates, so it is not correctly rounded)


@pytest.mark.parametrize("variant", [0, 1])
def test_blur_within_one_level_of_float_gaussian(variant):
    img = synth_frame(22, w=320, h=240)
    img[50:90, 60:120] = 255
    x = np.arange(-3, 4)
    k = np.exp(-x * x / 8.0); k /= k.sum()
    f = img.astype(np.float64)
    ref = ndi.correlate1d(ndi.correlate1d(f, k, axis=1, mode="mirror"), k, axis=0, mode="mirror")   # scipy "mirror" = BORDER_REFLECT_101
    got = O.gaussian_blur(img, variant).astype(np.float64)
    err = np.abs(got - np.minimum(ref, 255.0))
    assert err.max() <= (2.0 if variant == 0 else 3.0) and err.mean() < (0.5 if variant == 0 else 1.2)   # variant 1 gains (257 / 256)^2; 8-bit taps: {56,48,34,18}/256 resp. {55,49,34,18}/256 against .2161 .1907 .1311 .0702
    other = O.gaussian_blur(img, 1 - variant).astype(np.int32)
    d = np.abs(got.astype(np.int32) - other)
    assert d.max() <= 3 and d.any()                                 # the two OpenCV paths: up to (257 / 256)^2 - 1 = 0.8 % + rounding apart, and not identical


def test_fast_atan2_error_bound():
    rng = np.random.default_rng(3)
    y, x = rng.integers(-20000, 20000, 4000).astype(np.float32), rng.integers(-20000, 20000, 4000).astype(np.float32)
    got = np.array([O.fast_atan2(float(a), float(b)) for a, b in zip(y, x)])
    ref = np.degrees(np.arctan2(y.astype(np.float64), x.astype(np.float64))) % 360.0
    err = np.abs((got - ref + 180.0) % 360.0 - 180.0)
    assert err.max() < 0.3


def _pose_residuals(p, Xw, obs, w, K, T0):
    """p: SE3 increment (omega, upsilon) applied on the left of T0, as g2o's oplus; residuals sqrt(w) * (obs - project)."""
    from scipy.spatial.transform import Rotation as R
    Rm = R.from_rotvec(p[:3]).as_matrix() @ T0[0]
    # SE3 exp: t = V * upsilon; for a local check the first-order V = I + 0.5 [omega]x ... is not enough, so use the closed form
    th = np.linalg.norm(p[:3])
    Om = np.array([[0, -p[2], p[1]], [p[2], 0, -p[0]], [-p[1], p[0], 0]])
    V = np.eye(3) + (0.5 * Om + Om @ Om / 6.0 if th < 1e-5 else (1 - np.cos(th)) / th ** 2 * Om + (th - np.sin(th)) / th ** 3 * Om @ Om)
    t = R.from_rotvec(p[:3]).as_matrix() @ T0[1] + V @ p[3:]
    Xc = Xw @ Rm.T + t
    proj = np.stack([K[0] * Xc[:, 0] / Xc[:, 2] + K[2], K[1] * Xc[:, 1] / Xc[:, 2] + K[3]], 1)
    return ((obs - proj) * np.sqrt(w)[:, None]).ravel()


@pytest.mark.parametrize("seed", [100, 101, 102])
def test_pose_optimization_is_a_minimum_for_scipy(seed):
    from scipy.spatial.transform import Rotation as R
    pr = pose_problem(seed, 300, 0.1)
    n_good, T, outl = O.pose_optimization(pr["Xw"], pr["obs"], pr["inv_sigma2"], pr["K"], pr["T0"])
    inl = ~outl.astype(bool)
    assert inl.sum() == n_good
    q, t = T[:4].astype(np.float64), T[4:7].astype(np.float64)          # (qx, qy, qz, qw | t) as the C ABI carries poses
    T0 = (R.from_quat(q).as_matrix(), t)
    Xw, obs, w, K = pr["Xw"][inl].astype(np.float64), pr["obs"][inl].astype(np.float64), pr["inv_sigma2"][inl].astype(np.float64), pr["K"].astype(np.float64)
    f0 = _pose_residuals(np.zeros(6), Xw, obs, w, K, T0)
    sol = sopt.least_squares(_pose_residuals, np.zeros(6), args=(Xw, obs, w, K, T0), method="lm", xtol=1e-14, ftol=1e-14)
    c0, c1 = 0.5 * f0 @ f0, sol.cost
    # the pose comes back as float32 (Frame::SetPose): that rounding alone moves the cost by ~1e-6 relative
    assert c1 <= c0 * (1 + 1e-9) and (c0 - c1) <= 2e-6 * c0, (c0, c1)


def test_local_ba_cost_against_scipy_huber():
    b = ba_problem(seed=3, n_opt=3, n_fixed=2, n_points=60)
    a = (b["kf_pose"], b["kf_fixed"], b["mp_pos"], b["e_mp"], b["e_kf"], b["e_obs"], b["e_w"], b["K"])
    from scipy.spatial.transform import Rotation as R
    kf0, mp0 = b["kf_pose"].astype(np.float64), b["mp_pos"].astype(np.float64)
    fixed = b["kf_fixed"].astype(bool)
    opt_idx = np.nonzero(~fixed)[0]
    e_mp, e_kf, e_obs, e_w, K = b["e_mp"], b["e_kf"], b["e_obs"].astype(np.float64), b["e_w"].astype(np.float64), b["K"].astype(np.float64)
    delta = np.sqrt(5.991)

    def unpack(x):
        kf = kf0.copy()
        for j, k in enumerate(opt_idx):
            d = x[6 * j:6 * j + 6]
            Rk = R.from_rotvec(d[:3]).as_matrix() @ R.from_quat(kf0[k, :4]).as_matrix()
            kf[k, :4] = R.from_matrix(Rk).as_quat(); kf[k, 4:7] = R.from_rotvec(d[:3]).as_matrix() @ kf0[k, 4:7] + d[3:]
        return kf, x[6 * len(opt_idx):].reshape(-1, 3)

    def resid(x, kfmp=None):
        kf, mp = unpack(x) if kfmp is None else kfmp
        Rm = R.from_quat(kf[e_kf, :4]).as_matrix()
        Xc = np.einsum("eij,ej->ei", Rm, mp[e_mp]) + kf[e_kf, 4:7]
        proj = np.stack([K[0] * Xc[:, 0] / Xc[:, 2] + K[2], K[1] * Xc[:, 1] / Xc[:, 2] + K[3]], 1)
        return (e_obs - proj) * np.sqrt(e_w)[:, None]

    def huber_cost(r2):                      # g2o RobustKernelHuber: rho = e2 (e2 <= delta^2) else 2 delta sqrt(e2) - delta^2
        return float(np.sum(np.where(r2 <= delta * delta, r2, 2 * delta * np.sqrt(r2) - delta * delta)))

    x0 = np.concatenate([np.zeros(6 * len(opt_idx)), mp0.ravel()])
    c_start = huber_cost(np.sum(resid(x0) ** 2, 1))
    its, kf1, mp1, _ = O.local_ba(*a)
    c_oracle = huber_cost(np.sum(resid(None, (kf1.astype(np.float64), mp1.astype(np.float64))) ** 2, 1))
    assert c_oracle < 0.9 * c_start                                   # the 10-iteration LM does lower the robust cost

    # an independent solver started FROM the oracle's end state must not find the robust cost much lower: g2o's <= 10 LM iterations end within
    # 2 % of the minimum scipy reaches from there (started from the perturbed scene scipy's trust region stalls above the oracle's cost)
    kf0[:] = kf1.astype(np.float64)

    def fun(x):                                                       # per-edge robustified residual norm: sum(fun^2) = Huber cost
        r2 = np.sum(resid(x) ** 2, 1)
        return np.sqrt(np.where(r2 <= delta * delta, r2, 2 * delta * np.sqrt(r2) - delta * delta))
    x1 = np.concatenate([np.zeros(6 * len(opt_idx)), mp1.astype(np.float64).ravel()])
    assert abs(float(np.sum(fun(x1) ** 2)) - c_oracle) <= 1e-9 * c_oracle
    sol = sopt.least_squares(fun, x1, method="trf", xtol=1e-12, ftol=1e-12, max_nfev=100)
    c_scipy = 2 * sol.cost
    assert c_scipy <= c_oracle * (1 + 1e-9) and c_oracle <= c_scipy * 1.02, (c_start, c_oracle, c_scipy)


def _fast_bruteforce(img, threshold):
    """FAST-9/16 straight from its definition (Rosten & Drummond; the contract of cv::FAST with nonmaxSuppression = true), written without looking at
    how the oracle or the kernel evaluate it: p is a corner at t iff 9 CONTIGUOUS pixels of the 16-pixel Bresenham circle are all brighter than
    I(p) + t or all darker than I(p) - t; its score is the largest t for which it still is one; a corner survives iff its score is strictly
    greater than the scores of its 8 neighbours; a 3-pixel margin is never examined."""
    ring = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
    h, w = img.shape
    I = img.astype(np.int32)
    score = np.zeros((h, w), np.int32)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            d = np.array([I[y + dy, x + dx] - I[y, x] for dx, dy in ring])
            best = -1
            for sign in (1, -1):
                e = sign * d                                    # contrast on this side
                for s in range(16):
                    arc = min(e[(s + k) % 16] for k in range(9))
                    best = max(best, arc - 1)                    # all nine exceed t  <=>  t <= arc - 1
            if best >= threshold:
                score[y, x] = best
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            s = score[y, x]
            if s > 0 and all(s > score[y + j, x + i] for j in (-1, 0, 1) for i in (-1, 0, 1) if (i, j) != (0, 0)):
                out.append((x, y, s))
    return out


@pytest.mark.parametrize("seed,threshold", [(0, 20), (1, 7), (2, 20), (3, 12)])
def test_fast_against_the_definition(seed, threshold):
    """The oracle's cv::FAST restatement (oracle/orb_oracle.cc, which the GPU kernel is held bit-exact to) against a brute-force evaluation of the
    published definition on random textured patches: same corners, same scores, same non-maximum suppression, same order (row-major)."""
    rng = np.random.default_rng(seed)
    h, w = 34, 41
    base = rng.integers(0, 256, (h // 4 + 2, w // 4 + 2)).astype(np.float64)
    img = np.kron(base, np.ones((4, 4)))[:h, :w] + rng.normal(0, 12, (h, w))
    img = np.clip(img, 0, 255).astype(np.uint8)
    got = O.fast_cell(img, threshold)
    ref = _fast_bruteforce(img, threshold)
    assert len(ref) > 5, "the patch is supposed to have corners"
    assert [(int(k["x"]), int(k["y"]), int(k["response"])) for k in got] == ref


def test_undistort_points_inverts_the_published_distortion_model():
    """oracle/frame_oracle.cc restates cv::undistortPoints (not in the tree: parity unpinned).  Independent check with numpy: pushing the undistorted
    points through the FORWARD radial-tangential model (the one cv::projectPoints documents) must give the original pixels back, up to what five
    fixed-point iterations leave (a few hundredths of a pixel in the image corners for EuRoC's k1 = -0.28)."""
    K = np.array([458.654, 457.296, 367.215, 248.375], np.float32)
    d = np.array([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0], np.float32)
    rng = np.random.default_rng(0)
    xy = np.stack([rng.uniform(0, 752, 4000), rng.uniform(0, 480, 4000)], 1).astype(np.float32)
    xy[:4] = [[0, 0], [752, 0], [0, 480], [752, 480]]
    u = O.undistort_points(xy, K, d).astype(np.float64)
    fx, fy, cx, cy = K.astype(np.float64)
    k1, k2, p1, p2, k3 = d.astype(np.float64)
    x, y = (u[:, 0] - cx) / fx, (u[:, 1] - cy) / fy
    r2 = x * x + y * y
    rad = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
    xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    back = np.stack([fx * xd + cx, fy * yd + cy], 1)
    err = np.linalg.norm(back - xy.astype(np.float64), axis=1)
    inner = (np.abs(xy[:, 0] - 376) < 250) & (np.abs(xy[:, 1] - 240) < 160)
    assert err[inner].max() < 0.08, err[inner].max()          # (0.04 px measured: five iterations, k1 = -0.28)
    assert err.max() < 1.5, err.max()                          # the corners: five iterations stop short, as they do in OpenCV
    b = O.image_bounds(752, 480, K, d)
    assert b[0] < -100 and b[1] < -60 and b[2] > 850 and b[3] > 540, b
    assert np.array_equal(O.image_bounds(752, 480, K, np.zeros(5, np.float32)), np.array([0, 0, 752, 480], np.float32))
