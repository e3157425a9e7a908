"""rumi_track_frame (include/rumi_track.h): TrackWithMotionModel + TrackLocalMap data path in one device-resident call, against the ORACLE chain
run stage by stage on the same inputs: extract -> SearchByProjection(Cur, Last) (retry at 2 th) -> PoseOptimization -> outliers leave the frame
-> isInFrustum of the local points not seen in this frame -> SearchByProjection(Cur, local points) -> PoseOptimization.  Match indices, flags and
counts must be identical, poses within 1e-4 relative.  The scene is the closed-loop plane sequence of test_tracking_loop_gpu.py; the map holds the
frame-0 points (seen by the last frame) plus extra plane points only the local map knows, so the second search has something to add."""
import numpy as np
import pytest

import oracle_lib as O
from rumi_slam_amd.synth import synth_frame, warp_homography
from scene import K_TUM3
from test_tracking_loop_gpu import PLANE_D, _homography, _pose_gt

pytestmark = pytest.mark.gpu
RTOL = 1e-4
W, H = 640, 480


def _pose_close(T, Tref, what):
    assert np.linalg.norm(T[:4] - Tref[:4]) <= RTOL and np.linalg.norm(T[4:] - Tref[4:]) <= RTOL * max(1e-2, np.linalg.norm(Tref[4:])), f"{what}: {T} vs {Tref}"


def _pose_matrices(T):
    """Frame::UpdatePoseMatrices in float32, operation by operation as k_track_pose_matrices / Eigen do it."""
    f = np.float32
    x, y, z, w = (f(v) for v in T[:4])
    tx, ty, tz = f(2) * x, f(2) * y, f(2) * z
    twx, twy, twz, txx, txy, txz, tyy, tyz, tzz = tx * w, ty * w, tz * w, tx * x, ty * x, tz * x, ty * y, tz * y, tz * z
    R = np.array([f(1) - (tyy + tzz), txy - twz, txz + twy, txy + twz, f(1) - (txx + tzz), tyz - twx, txz - twy, tyz + twx, f(1) - (txx + tyy)], np.float32)
    t = np.array(T[4:], np.float32)
    q = np.array([-x, -y, -z], np.float32); v = t * f(-1)
    u = np.cross(q, v).astype(np.float32); u = u + u
    c = np.cross(q, u).astype(np.float32)
    return R, t, ((v + w * u) + c).astype(np.float32)


def _apply_stale(fr, skip, discarded_ids, pts):
    """Discarded outliers whose mbTrackInView an earlier frame left set (a monocular frame's discard loop clears mbTrackInViewR only: Nleft = -1,
    Tracking.cc:2489-2508) are not re-projected but searched at their OLD projection (ORBmatcher.cc:46-60): pts["stale_in_view"] / ["stale_proj"]
    (mTrackProjX, mTrackProjY, mnTrackScaleLevel, mTrackViewCos, mTrackDepth).  Returns the in_view array the device reports (2 = such a point)."""
    in_view = fr["track_in_view"].copy()
    if pts.get("stale_in_view") is None:
        return in_view
    for j in np.unique(discarded_ids):
        if pts["local"][j] and not pts["bad"][j] and pts["stale_in_view"][j]:
            sp = pts["stale_proj"][j]
            skip[j] = 0
            fr["track_in_view"][j] = 1; fr["proj_x"][j] = sp[0]; fr["proj_y"][j] = sp[1]; fr["scale_level"][j] = int(sp[2]); fr["view_cos"][j] = sp[3]; fr["track_depth"][j] = sp[4]
            in_view[j] = 2
    return in_view


def _oracle_step(orc, img, sf, inv_sigma2, T_pred, last, pts, th_local):
    """The reference's sequence with the oracle's functions; returns everything rumi_track_frame reports."""
    mono, keys, desc = orc.extract(img)
    n = len(keys)
    r = dict(n=n, mono_index=mono, keys=keys, desc=desc)
    cur0 = np.full(n, -1, np.int32)
    a = (keys, desc, W, H, sf, T_pred, K_TUM3, last["keys"], last["mp"], last["outlier"], pts["pos"], pts["desc"], pts["obs"], cur0)
    th = 15.0
    nm, cur = O.search_by_projection_frame(*a, th, True)
    if nm < 20:
        th = 30.0
        nm, cur = O.search_by_projection_frame(*a, th, True)
    r.update(th_motion=int(th), nmatches_motion=nm)
    if nm < 20:                                                 # TrackWithMotionModel gives up (Tracking.cc:2476-2483): the frame keeps what it has
        r.update(ngood_motion=0, Tcw_motion=T_pred, nmatches_map=0, frame_mp_motion=cur, in_view=np.zeros(len(pts["obs"]), np.uint8), n_to_match=0,
                 nmatches_local=0, frame_mp=cur, ngood_local=0, Tcw=T_pred, outlier=np.zeros(n, np.uint8), matches_inliers=0)
        return r
    idx = np.nonzero(cur >= 0)[0]
    ng, T1, out = O.pose_optimization(pts["pos"][cur[idx]], np.stack([keys["x"][idx], keys["y"][idx]], 1), inv_sigma2[keys["octave"][idx]], K_TUM3, T_pred)
    seen = np.zeros(len(pts["obs"]), np.uint8)
    seen[cur[idx]] = 1
    inl = idx[out == 0]
    r.update(ngood_motion=ng, Tcw_motion=T1, nmatches_map=int((pts["obs"][cur[inl]] > 0).sum()))
    discarded_ids = cur[idx[out != 0]].copy()
    cur[idx[out != 0]] = -1
    cur[inl[pts["bad"][cur[inl]] != 0]] = -1
    r["frame_mp_motion"] = cur.copy()
    R, t, Ow = _pose_matrices(T1)
    skip = ((pts["local"] == 0) | (seen != 0) | (pts["bad"] != 0)).astype(np.uint8)
    log_sf = float(np.log(np.float32(1.2)))
    fr = O.is_in_frustum(R, t, Ow, K_TUM3, W, H, log_sf, 8, 0.5, pts)
    for k in fr:
        fr[k] = np.where(skip != 0, np.array(-1 if k in ("proj_x", "proj_y") else 0, fr[k].dtype), fr[k])
    n_to_match = int(fr["track_in_view"].sum())
    r.update(in_view=_apply_stale(fr, skip, discarded_ids, pts), n_to_match=n_to_match, Rcw=R, tcw=t, Ow=Ow)
    nml, cur2 = O.search_by_projection_mappoints(keys, desc, W, H, sf, dict(fr, is_bad=skip, desc=pts["desc"], obs=pts["obs"]), cur, th_local, False, 0.0, 0.8)
    idx2 = np.nonzero(cur2 >= 0)[0]
    ng2, T2, out2 = O.pose_optimization(pts["pos"][cur2[idx2]], np.stack([keys["x"][idx2], keys["y"][idx2]], 1), inv_sigma2[keys["octave"][idx2]], K_TUM3, T1)
    outl = np.zeros(n, np.uint8)
    outl[idx2] = out2
    r.update(nmatches_local=nml, frame_mp=cur2, ngood_local=ng2, Tcw=T2, outlier=outl,
             matches_inliers=int(((out2 == 0) & (pts["obs"][cur2[idx2]] > 0)).sum()))
    return r


@pytest.mark.parametrize("th_local", [1.0, 3.0])
def test_track_frame_equals_oracle_chain(th_local):
    from rumi_slam_amd.extractor import ORBextractor
    from rumi_slam_amd.tracker import Tracker
    trk = Tracker(1000, 1.2, 8, 20, 7, W, H, 4096)
    ext = ORBextractor(1000, 1.2, 8, 20, 7)
    orc = O.OracleExtractor(1000, 1.2, 8, 20, 7)
    sf, inv_sigma2 = ext.GetScaleFactors(), ext.GetInverseScaleSigmaSquares()
    img0 = synth_frame(4242)
    fx, fy, cx, cy = K_TUM3.astype(np.float64)
    _, keys0, desc0 = ext(img0)
    n0 = len(keys0)
    rng = np.random.default_rng(7)
    # map: the frame-0 points (half of them known to the last frame), all of them local; a few bad, a few never observed
    pos = np.stack([(keys0["x"] - cx) / fx * PLANE_D, (keys0["y"] - cy) / fy * PLANE_D, np.full(n0, PLANE_D)], 1).astype(np.float32)
    dist0 = np.linalg.norm(pos, axis=1).astype(np.float32)
    lvl = keys0["octave"]
    pts = dict(pos=pos, normal=(pos / dist0[:, None]).astype(np.float32),        # mean viewing direction: from the camera to the point
               max_dist=(dist0 * sf[lvl]).astype(np.float32), min_dist=(dist0 * sf[lvl] / sf[7]).astype(np.float32), desc=desc0.copy(),
               obs=np.where(rng.random(n0) < 0.05, 0, 1).astype(np.int32), bad=(rng.random(n0) < 0.03).astype(np.uint8), local=np.ones(n0, np.uint8))
    known = rng.random(n0) < 0.5
    last = dict(keys=keys0, mp=np.where(known, np.arange(n0), -1).astype(np.int32), outlier=np.zeros(n0, np.uint8))
    T = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    for t in range(1, 6):
        q_gt, t_gt = _pose_gt(t)
        img = warp_homography(img0, _homography(q_gt, t_gt))
        ref = _oracle_step(orc, img, sf, inv_sigma2, T, last, pts, th_local)
        got = trk.track(img, K_TUM3, T, last["keys"], last["mp"], last["outlier"], pts, 15.0, th_local)
        assert got["n"] == ref["n"] and got["mono_index"] == ref["mono_index"] and got["keys"].tobytes() == ref["keys"].tobytes() \
            and np.array_equal(got["desc"], ref["desc"]), f"frame {t}: extraction"
        for k in ("th_motion", "nmatches_motion", "ngood_motion", "nmatches_map", "n_to_match", "nmatches_local", "ngood_local", "matches_inliers"):
            assert got[k] == ref[k], f"frame {t}: {k} {got[k]} vs {ref[k]}"
        for k in ("frame_mp_motion", "in_view", "frame_mp", "outlier"):
            assert np.array_equal(got[k], ref[k]), f"frame {t}: {k}"
        for k in ("Rcw", "tcw", "Ow"):
            R, tt, Ow = _pose_matrices(got["Tcw_motion"])
            assert np.array_equal(got[k], dict(Rcw=R, tcw=tt, Ow=Ow)[k]), f"frame {t}: {k} is not UpdatePoseMatrices of the device's pose"
        _pose_close(got["Tcw_motion"], ref["Tcw_motion"], f"frame {t} pose after the motion model")
        _pose_close(got["Tcw"], ref["Tcw"], f"frame {t} pose after the local map")
        assert ref["nmatches_local"] > 50 and ref["matches_inliers"] > 100, "the local search is supposed to add matches"
        # hand over: the frame becomes the last frame (mvpMapPoints with the outliers of the LAST optimisation still flagged, mono: Tracking.cc:2587)
        last = dict(keys=got["keys"], mp=got["frame_mp"], outlier=got["outlier"])
        T = got["Tcw"]


def test_track_frame_lost_and_empty_inputs():
    from rumi_slam_amd.tracker import Tracker
    trk = Tracker(1000, 1.2, 8, 20, 7, W, H, 1024)
    img = synth_frame(99)
    none = dict(pos=np.zeros((0, 3), np.float32), normal=np.zeros((0, 3), np.float32), min_dist=np.zeros(0, np.float32), max_dist=np.zeros(0, np.float32),
                desc=np.zeros((0, 32), np.uint8), obs=np.zeros(0, np.int32), bad=np.zeros(0, np.uint8), local=np.zeros(0, np.uint8))
    from rumi_slam_amd.capi import KP_DTYPE
    T = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    got = trk.track(img, K_TUM3, T, np.zeros(0, KP_DTYPE), np.zeros(0, np.int32), np.zeros(0, np.uint8), none)
    assert got["n"] > 900 and got["nmatches_motion"] == 0 and got["ngood_motion"] == 0 and (got["frame_mp"] == -1).all() and np.array_equal(got["Tcw"], T)
    orc = O.OracleExtractor(1000, 1.2, 8, 20, 7)
    mono, keys, desc = orc.extract(img)
    assert got["keys"].tobytes() == keys.tobytes() and np.array_equal(got["desc"], desc)


def _scene(seed=4242, n_keep=1.0):
    from rumi_slam_amd.extractor import ORBextractor
    ext = ORBextractor(1000, 1.2, 8, 20, 7)
    sf, inv_sigma2 = ext.GetScaleFactors(), ext.GetInverseScaleSigmaSquares()
    img0 = synth_frame(seed)
    fx, fy, cx, cy = K_TUM3.astype(np.float64)
    _, keys0, desc0 = ext(img0)
    n0 = len(keys0)
    pos = np.stack([(keys0["x"] - cx) / fx * PLANE_D, (keys0["y"] - cy) / fy * PLANE_D, np.full(n0, PLANE_D)], 1).astype(np.float32)
    dist0 = np.linalg.norm(pos, axis=1).astype(np.float32)
    lvl = keys0["octave"]
    pts = dict(pos=pos, normal=(pos / dist0[:, None]).astype(np.float32), max_dist=(dist0 * sf[lvl]).astype(np.float32),
               min_dist=(dist0 * sf[lvl] / sf[7]).astype(np.float32), desc=desc0.copy(), obs=np.ones(n0, np.int32), bad=np.zeros(n0, np.uint8),
               local=np.ones(n0, np.uint8))
    rng = np.random.default_rng(3)
    last = dict(keys=keys0, mp=np.where(rng.random(n0) < n_keep, np.arange(n0), -1).astype(np.int32), outlier=np.zeros(n0, np.uint8))
    return img0, sf, inv_sigma2, pts, last


def _compare(got, ref, what):
    assert got["n"] == ref["n"] and got["keys"].tobytes() == ref["keys"].tobytes() and np.array_equal(got["desc"], ref["desc"]), f"{what}: extraction"
    for k in ("th_motion", "nmatches_motion", "ngood_motion", "nmatches_map", "n_to_match", "nmatches_local", "ngood_local", "matches_inliers"):
        assert got[k] == ref[k], f"{what}: {k} {got[k]} vs {ref[k]}"
    for k in ("frame_mp_motion", "in_view", "frame_mp", "outlier"):
        assert np.array_equal(got[k], ref[k]), f"{what}: {k}"
    _pose_close(got["Tcw_motion"], np.asarray(ref["Tcw_motion"], np.float32), what + " pose after the motion model")
    _pose_close(got["Tcw"], np.asarray(ref["Tcw"], np.float32), what + " pose after the local map")


def test_track_frame_retry_at_twice_the_radius_and_giving_up():
    """A poor prediction: the search at th = 15 finds fewer than 20 matches and is repeated at 30 (Tracking.cc:2469-2474); a hopeless one: still fewer
    than 20, TrackWithMotionModel gives up, no optimisation runs, the frame keeps the few matches and the predicted pose."""
    from rumi_slam_amd.tracker import Tracker
    trk = Tracker(1000, 1.2, 8, 20, 7, W, H, 4096)
    orc = O.OracleExtractor(1000, 1.2, 8, 20, 7)
    img0, sf, inv_sigma2, pts, last = _scene(n_keep=0.08)     # few map points in the last frame: the match count sits near the threshold
    q_gt, t_gt = _pose_gt(2)
    img = warp_homography(img0, _homography(q_gt, t_gt))
    seen = set()
    for shift in (0.0, 0.05, 0.08, 0.11, 0.5, 1.5, 6.0):       # prediction off by a growing sideways translation (the last ones: nothing left to match)
        T = np.array([0, 0, 0, 1, shift, 0, 0], np.float32)
        ref = _oracle_step(orc, img, sf, inv_sigma2, T, last, pts, 1.0)
        got = trk.track(img, K_TUM3, T, last["keys"], last["mp"], last["outlier"], pts, 15.0, 1.0)
        _compare(got, ref, f"shift {shift}")
        seen.add((ref["th_motion"], ref["nmatches_motion"] >= 20))
    assert (30, True) in seen or (30, False) in seen, f"no case exercised the retry: {seen}"
    assert any(not ok for _, ok in seen), f"no case gave up: {seen}"


def test_track_frame_strided_image_and_bad_points():
    """The image as a view into a wider buffer (stride != width); a third of the map points bad, some unobserved: the local search must leave them alone."""
    from rumi_slam_amd.tracker import Tracker
    trk = Tracker(1000, 1.2, 8, 20, 7, W, H, 4096)
    orc = O.OracleExtractor(1000, 1.2, 8, 20, 7)
    img0, sf, inv_sigma2, pts, last = _scene(n_keep=0.6)
    rng = np.random.default_rng(11)
    pts["bad"] = (rng.random(len(pts["obs"])) < 0.33).astype(np.uint8)
    pts["obs"] = np.where(rng.random(len(pts["obs"])) < 0.2, 0, 2).astype(np.int32)
    pts["local"] = (rng.random(len(pts["obs"])) < 0.8).astype(np.uint8)
    q_gt, t_gt = _pose_gt(1)
    img = warp_homography(img0, _homography(q_gt, t_gt))
    wide = np.zeros((H, W + 24), np.uint8)
    wide[:, 8:8 + W] = img
    view = wide[:, 8:8 + W]
    assert not view.flags["C_CONTIGUOUS"]
    T = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    ref = _oracle_step(orc, img, sf, inv_sigma2, T, last, pts, 1.0)
    got = trk.track(view, K_TUM3, T, last["keys"], last["mp"], last["outlier"], pts, 15.0, 1.0)
    _compare(got, ref, "strided image")
    assert ref["nmatches_local"] > 20


def test_single_queue_step_equals_stage_by_stage(tmp_path):
    """rumi_track_frame queues its usual case (>= 20 matches, no list overflow) in one go and falls back to the stage-by-stage path otherwise.
    Both paths must return the same step: feature budgets 500 / 1000 / 2000 (the last beyond the LDS form of the pose kernel), last frames with
    many, fewer than 20 and no known points (tools/track_spec_check.py, once per path: the switch is read once per process)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dumps = []
    for spec in ("0", "1"):
        out = str(tmp_path / f"spec{spec}.npz")
        env = dict(os.environ, RUMI_TRACK_SPECULATE=spec)
        subprocess.run([sys.executable, os.path.join(root, "tools", "track_spec_check.py"), out], check=True, env=env, timeout=600)
        dumps.append(np.load(out))
    a, b = dumps
    assert set(a.files) == set(b.files) and len(a.files) > 300
    bad = [k for k in a.files if not np.array_equal(a[k], b[k])]
    assert not bad, f"the two paths differ in {bad[:8]}"
    assert any(int(a[k]) < 20 for k in a.files if k.endswith("_nmatches_motion")) and any(int(a[k]) >= 20 for k in a.files if k.endswith("_nmatches_motion"))


@pytest.mark.gpu
def test_frame_captured_into_the_trackers_pinned_buffer():
    """rumi_track_image_buffer: a frame written into the tracker's pinned staging memory and passed to rumi_track_frame / rumi_track_extract by
    that pointer skips the staging copy; the step must be the one a pageable copy of the frame gives."""
    from rumi_slam_amd.tracker import Tracker
    from rumi_slam_amd.synth import synth_frame
    trk = Tracker(1000, 1.2, 8, 20, 7, 640, 480, 4096)
    img = synth_frame(777)
    mono_a, keys_a, desc_a = trk.extract(img)
    buf = trk.image_buffer(640, 480)
    assert buf.shape == (480, 640) and buf.strides == (640, 1)
    buf[:] = img
    mono_b, keys_b, desc_b = trk.extract(buf)
    assert mono_a == mono_b and keys_a.tobytes() == keys_b.tobytes() and np.array_equal(desc_a, desc_b) and len(keys_a) > 500
    r_w, r_h = trk.image_buffer(322, 240).shape[1], trk.image_buffer(322, 240).strides[0]
    assert (r_w, r_h) == (322, 324)                              # a narrower frame: rows padded to 4 bytes
