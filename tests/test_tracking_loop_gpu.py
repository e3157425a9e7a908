"""Closed-loop synthetic tracking — the stated substitute for BASELINE.json config 1 (TUM fr3 on the Tracking thread), which
cannot run here: no dataset, no ORBvoc.txt, no OpenCV.

A textured plane 4 m in front of camera 0 is viewed by a camera on a smooth trajectory; frame t is frame 0 under the exact
plane homography K (R + t n^T / d) K^-1, so the true pose of every frame is known.  The GPU chain runs the reference's
TrackWithMotionModel sequence per frame — ORBextractor -> SearchByProjection(Cur, Last, th=15, retry 30) -> PoseOptimization
-> drop outliers (Tracking.cc:2434-2530) — feeding each frame's result into the next.  At every step the oracle is evaluated on
the SAME inputs the GPU chain saw: key-points/descriptors and match indices must be bit-identical, poses within 1e-4 relative,
outlier flags identical.  Finally the estimated trajectory is compared with the ground truth (an ATE-style sanity bound)."""
import numpy as np
import pytest

import oracle_lib as O
from rumi_slam_amd.synth import synth_frame, warp_homography
from scene import K_TUM3, quat_from_rotvec, quat_rotate

pytestmark = pytest.mark.gpu
RTOL = 1e-4
N_FRAMES = 12
PLANE_D = 4.0


def _pose_gt(t):
    rv = np.array([0.0015, -0.0025, 0.004]) * t
    tr = np.array([0.012, -0.006, 0.010]) * t
    return quat_from_rotvec(rv), tr


def _homography(q, tr):
    fx, fy, cx, cy = K_TUM3.astype(np.float64)
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]])
    _, R = quat_rotate(q, np.zeros((1, 3)))
    return K @ (R + np.outer(tr, [0, 0, 1]) / PLANE_D) @ np.linalg.inv(K)


def test_closed_loop_tracking_matches_oracle_and_ground_truth():
    from rumi_slam_amd.extractor import ORBextractor
    from rumi_slam_amd.matcher import FrameView, ORBmatcher
    from rumi_slam_amd.optimizer import Optimizer
    ext = ORBextractor(1000, 1.2, 8, 20, 7)
    orc = O.OracleExtractor(1000, 1.2, 8, 20, 7)
    matcher, opt = ORBmatcher(0.9, True), Optimizer()
    sf = ext.GetScaleFactors()
    inv_sigma2 = ext.GetInverseScaleSigmaSquares()
    w, h = 640, 480
    img0 = synth_frame(4242)
    fx, fy, cx, cy = K_TUM3.astype(np.float64)

    # frame 0: map initialised from its key-points on the plane (world = camera 0)
    _, keys, desc = ext(img0)
    assert keys.tobytes() == orc.extract(img0)[1].tobytes()
    n0 = len(keys)
    mp_pos = np.stack([(keys["x"] - cx) / fx * PLANE_D, (keys["y"] - cy) / fy * PLANE_D, np.full(n0, PLANE_D)], 1).astype(np.float32)
    mp_desc = desc.copy()
    mp_obs = np.ones(n0, np.int32)
    last = dict(keys=keys, mp=np.arange(n0, dtype=np.int32), outlier=np.zeros(n0, np.uint8))
    T = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    errs_t, errs_r, tracked = [], [], []
    for t in range(1, N_FRAMES):
        q_gt, t_gt = _pose_gt(t)
        img = warp_homography(img0, _homography(q_gt, t_gt))
        mono, keys, desc = ext(img)
        om, ok, od = orc.extract(img)
        assert mono == om and keys.tobytes() == ok.tobytes() and np.array_equal(desc, od), f"frame {t}: extraction differs"
        F = FrameView(keys, desc, w, h, sf)
        cur0 = np.full(F.n, -1, np.int32)
        th = 15.0
        args = (T, K_TUM3, last["keys"], last["mp"], last["outlier"], mp_pos, mp_desc, mp_obs, cur0)
        nm, cur_mp = matcher.SearchByProjection_Frame(F, *args, th)
        nm_ref, cur_ref = O.search_by_projection_frame(keys, desc, w, h, sf, T, K_TUM3, last["keys"], last["mp"], last["outlier"], mp_pos,
                                                       mp_desc, mp_obs, cur0, th, True)
        if nm < 20:                                                     # Tracking.cc:2469-2474
            th = 30.0
            nm, cur_mp = matcher.SearchByProjection_Frame(F, *args, th)
            nm_ref, cur_ref = O.search_by_projection_frame(keys, desc, w, h, sf, T, K_TUM3, last["keys"], last["mp"], last["outlier"], mp_pos,
                                                           mp_desc, mp_obs, cur0, th, True)
        assert nm == nm_ref and np.array_equal(cur_mp, cur_ref), f"frame {t}: match indices differ"
        assert nm >= 100, f"frame {t}: only {nm} matches"
        idx = np.nonzero(cur_mp >= 0)[0]
        Xw = mp_pos[cur_mp[idx]]
        obs = np.stack([keys["x"][idx], keys["y"][idx]], 1)
        wgt = inv_sigma2[keys["octave"][idx]]
        ng, Tn, out = opt.PoseOptimization(Xw, obs, wgt, K_TUM3, T)
        ng_ref, T_ref, out_ref = O.pose_optimization(Xw, obs, wgt, K_TUM3, T)
        assert ng == ng_ref and np.array_equal(out, out_ref), f"frame {t}: inliers {ng} vs {ng_ref}"
        assert np.linalg.norm(Tn[:4] - T_ref[:4]) <= RTOL and np.linalg.norm(Tn[4:] - T_ref[4:]) <= RTOL * max(1e-2, np.linalg.norm(T_ref[4:])), \
            f"frame {t}: pose {Tn} vs oracle {T_ref}"
        # discard outliers, hand the frame over (Tracking.cc:2489-2513)
        cur_mp[idx[out != 0]] = -1
        outl = np.zeros(F.n, np.uint8)
        last = dict(keys=keys, mp=cur_mp, outlier=outl)
        T = Tn
        tracked.append(ng)
        dq = np.abs(np.dot(Tn[:4].astype(np.float64), q_gt))
        errs_r.append(2 * np.degrees(np.arccos(min(1.0, dq))))
        errs_t.append(np.linalg.norm(Tn[4:].astype(np.float64) - t_gt))
    rmse_t = float(np.sqrt(np.mean(np.square(errs_t))))
    print(f"closed loop: {N_FRAMES - 1} frames, inliers {min(tracked)}..{max(tracked)}, translation RMSE {rmse_t * 1e3:.2f} mm, "
          f"max rotation error {max(errs_r):.4f} deg")
    assert rmse_t < 0.02 and max(errs_r) < 0.3, (errs_t, errs_r)


def test_two_threads_two_handles():
    """The reference runs one extractor in the Tracking thread and another inside KFDSample, matchers and optimisers on several threads
    (SURVEY §8b): handles are per thread, the library keeps no shared mutable state.  Two Python threads (ctypes releases the GIL) hammer
    their own extractor / matcher / optimiser concurrently; every result must equal the single-threaded one."""
    import threading
    from ba_scene import pose_problem
    from rumi_slam_amd.extractor import ORBextractor
    from rumi_slam_amd.matcher import FrameView, ORBmatcher
    from rumi_slam_amd.optimizer import Optimizer
    from rumi_slam_amd.synth import synth_frame
    from scene import TrackingScene
    imgs = [synth_frame(4000 + i) for i in range(4)]
    s = TrackingScene(1)
    p = pose_problem(9, 250, 0.1)
    mpv = s.mappoint_view()                                  # draws from the scene's RNG: once, shared by every call

    def work(out, reps):
        ext, m, opt = ORBextractor(1000, 1.2, 8, 20, 7), ORBmatcher(0.8, True), Optimizer()
        F = FrameView(s.cur_keys, s.cur_desc, s.w, s.h, s.sf)
        res = []
        for r in range(reps):
            mono, k, d = ext(imgs[r % 4])
            n1, fm = m.SearchByProjection_MapPoints(F, mpv, np.full(F.n, -1, np.int32), 3.0)
            ng, T, o = opt.PoseOptimization(p["Xw"], p["obs"], p["inv_sigma2"], p["K"], p["T0"])
            res.append((mono, k.tobytes(), d.tobytes(), n1, fm.tobytes(), ng, T.tobytes(), o.tobytes()))
        out.append(res)
    ref = []
    work(ref, 4)
    outs = [[], []]
    th = [threading.Thread(target=work, args=(outs[i], 12)) for i in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    for o in outs:
        assert len(o) == 1 and len(o[0]) == 12
        for r, got in enumerate(o[0]):
            names = ("mono", "key-points", "descriptors", "matches", "frame_mp", "pose inliers", "pose", "outlier flags")
            bad = [names[i] for i in range(8) if got[i] != ref[0][r % 4][i]]
            assert not bad, f"thread result {r} differs from the single-threaded one in {bad}"
