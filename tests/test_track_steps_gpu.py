"""The step-wise entries of include/rumi_track.h -- rumi_track_extract, rumi_track_motion, rumi_track_reference_keyframe, rumi_track_local: one
member function of Tracking per call, the frame resident on the device in between -- against the ORACLE chain of the same functions
(R/lib_src/Tracking.cc:2441-2518, 2324-2375, 2520-2607, 2996-3055), and against the fused rumi_track_frame where the two must agree.
Match indices, flags and counts identical; poses within 1e-4 relative."""
import numpy as np
import pytest

import oracle_lib as O
from rumi_slam_amd.synth import synth_frame, warp_homography
from scene import K_TUM3
from test_track_frame_gpu import H, W, _apply_stale, _oracle_step, _pose_close, _pose_matrices, _scene
from test_tracking_loop_gpu import _homography, _pose_gt
from voc_scene import synthetic_vocabulary

pytestmark = pytest.mark.gpu


def _oracle_motion(keys, desc, sf, inv_sigma2, T_pred, last, pts, bounds=None):
    """Tracking::TrackWithMotionModel alone: the frame's vector when the function returns (bad points still in it: their removal is
    SearchLocalPoints' first loop), the discarded outliers, the counts."""
    n = len(keys)
    cur0 = np.full(n, -1, np.int32)
    a = (keys, desc, W if bounds is None else bounds, H, sf, T_pred, K_TUM3, last["keys"], last["mp"], last["outlier"], pts["pos"], pts["desc"], pts["obs"], cur0)
    th = 15.0
    nm, cur = O.search_by_projection_frame(*a, th, True)
    if nm < 20:
        th = 30.0
        nm, cur = O.search_by_projection_frame(*a, th, True)
    r = dict(th_motion=int(th), nmatches_motion=nm, frame_mp=cur.copy(), discarded=np.full(n, -1, np.int32), ngood_motion=0, nmatches_map=0, Tcw_motion=T_pred)
    if nm < 20:
        return r
    idx = np.nonzero(cur >= 0)[0]
    ng, T1, out = O.pose_optimization(pts["pos"][cur[idx]], np.stack([keys["x"][idx], keys["y"][idx]], 1), inv_sigma2[keys["octave"][idx]], K_TUM3, T_pred)
    inl = idx[out == 0]
    r["discarded"][idx[out != 0]] = cur[idx[out != 0]]
    r.update(ngood_motion=ng, Tcw_motion=T1, nmatches_map=int((pts["obs"][cur[inl]] > 0).sum()))
    cur[idx[out != 0]] = -1
    r["frame_mp"] = cur
    return r


def _oracle_local(keys, desc, sf, inv_sigma2, T1, frame_mp_in, seen_in, pts, th_local, bounds=None):
    """Tracking::TrackLocalMap after UpdateLocalMap: SearchLocalPoints' two loops, the search, PoseOptimization, the statistics loop."""
    n, nmp = len(keys), len(pts["obs"])
    cur = frame_mp_in.copy()
    seen = seen_in.copy()
    for i in range(n):
        if cur[i] >= 0:
            if pts["bad"][cur[i]]:
                cur[i] = -1
            else:
                seen[cur[i]] = 1
    R, t, Ow = _pose_matrices(T1)
    skip = ((pts["local"] == 0) | (seen != 0) | (pts["bad"] != 0)).astype(np.uint8)
    fr = O.is_in_frustum(R, t, Ow, K_TUM3, W if bounds is None else bounds, H, float(np.log(np.float32(1.2))), 8, 0.5, pts)
    for k in fr:
        fr[k] = np.where(skip != 0, np.array(-1 if k in ("proj_x", "proj_y") else 0, fr[k].dtype), fr[k])
    n_to_match = int(fr["track_in_view"].sum())
    in_view = _apply_stale(fr, skip, np.nonzero(seen_in)[0], pts)
    nml, cur2 = O.search_by_projection_mappoints(keys, desc, W if bounds is None else bounds, H, sf, dict(fr, is_bad=skip, desc=pts["desc"], obs=pts["obs"]), cur, th_local, False, 0.0, 0.8)
    idx2 = np.nonzero(cur2 >= 0)[0]
    ng2, T2, out2 = O.pose_optimization(pts["pos"][cur2[idx2]], np.stack([keys["x"][idx2], keys["y"][idx2]], 1), inv_sigma2[keys["octave"][idx2]], K_TUM3, T1)
    outl = np.zeros(n, np.uint8)
    outl[idx2] = out2
    return dict(in_view=in_view, n_to_match=n_to_match, nmatches_local=nml, frame_mp=cur2, ngood_local=ng2, Tcw=T2, outlier=outl,
                matches_inliers=int(((out2 == 0) & (pts["obs"][cur2[idx2]] > 0)).sum()), Rcw=R, tcw=t, Ow=Ow)


def _seen_from(discarded, nmp):
    s = np.zeros(nmp, np.uint8)
    s[discarded[discarded >= 0]] = 1
    return s


@pytest.mark.parametrize("th_local", [1.0, 3.0])
def test_motion_then_local_equal_the_oracle_chain_and_the_fused_call(th_local):
    from rumi_slam_amd.tracker import Tracker
    fused, steps = Tracker(1000, 1.2, 8, 20, 7, W, H, 4096), Tracker(1000, 1.2, 8, 20, 7, W, H, 4096)
    orc = O.OracleExtractor(1000, 1.2, 8, 20, 7)
    img0, sf, inv_sigma2, pts, last = _scene(n_keep=0.5)
    rng = np.random.default_rng(7)
    n0 = len(pts["obs"])
    pts["obs"] = np.where(rng.random(n0) < 0.05, 0, 1).astype(np.int32)
    pts["bad"] = (rng.random(n0) < 0.03).astype(np.uint8)
    T = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    for t in range(1, 5):
        q_gt, t_gt = _pose_gt(t)
        img = warp_homography(img0, _homography(q_gt, t_gt))
        mono, keys, desc = steps.extract(img)
        omono, okeys, odesc = orc.extract(img)
        assert mono == omono and keys.tobytes() == okeys.tobytes() and np.array_equal(desc, odesc), f"frame {t}: extraction"
        # --- TrackWithMotionModel
        rm = _oracle_motion(okeys, odesc, sf, inv_sigma2, T, last, pts)
        gm = steps.motion(K_TUM3, T, last["keys"], last["mp"], last["outlier"], pts)
        for k in ("th_motion", "nmatches_motion", "ngood_motion", "nmatches_map"):
            assert gm[k] == rm[k], f"frame {t} motion: {k} {gm[k]} vs {rm[k]}"
        assert np.array_equal(gm["frame_mp"], rm["frame_mp"]) and np.array_equal(gm["discarded"], rm["discarded"]), f"frame {t} motion: vectors"
        assert (rm["discarded"] >= 0).sum() > 0 or t > 1
        _pose_close(gm["Tcw_motion"], rm["Tcw_motion"], f"frame {t} pose after the motion model")
        # --- (the host's UpdateLocalMap would run here) --- TrackLocalMap
        seen = _seen_from(gm["discarded"], n0)
        rl = _oracle_local(okeys, odesc, sf, inv_sigma2, gm["Tcw_motion"], gm["frame_mp"], seen, pts, th_local)
        gl = steps.local(K_TUM3, gm["Tcw_motion"], gm["frame_mp"], pts, seen, th_local)
        for k in ("n_to_match", "nmatches_local", "ngood_local", "matches_inliers"):
            assert gl[k] == rl[k], f"frame {t} local: {k} {gl[k]} vs {rl[k]}"
        for k in ("frame_mp", "outlier", "in_view", "Rcw", "tcw", "Ow"):
            assert np.array_equal(gl[k], rl[k]), f"frame {t} local: {k}"
        _pose_close(gl["Tcw"], rl["Tcw"], f"frame {t} pose after the local map")
        # --- the fused call on the same inputs ends in the same frame (same table, the discarded points marked seen)
        gf = fused.track(img, K_TUM3, T, last["keys"], last["mp"], last["outlier"], pts, 15.0, th_local)
        for k in ("frame_mp", "outlier", "in_view"):
            assert np.array_equal(gf[k], gl[k]), f"frame {t}: fused vs step-wise {k}"
        assert gf["matches_inliers"] == gl["matches_inliers"] and gf["nmatches_local"] == gl["nmatches_local"]
        assert np.array_equal(gf["Tcw"], gl["Tcw"]), "same kernels on the same inputs: the same pose, bit for bit"
        last = dict(keys=keys, mp=gl["frame_mp"], outlier=gl["outlier"])
        T = gl["Tcw"]


def test_motion_model_gives_up_then_reference_keyframe_takes_over():
    """A hopeless prediction: TrackWithMotionModel finds fewer than 20 matches and returns before optimising (the frame's vector then holds the
    search's result and NOTHING else has happened: no discard list).  Tracking::Track falls back to TrackReferenceKeyFrame on the SAME
    resident frame: BoW transform and FeatureVector on the device, SearchByBoW against the key-frame, PoseOptimization, discard."""
    from rumi_slam_amd.matcher import FeatureVector, FrameView
    from rumi_slam_amd.tracker import Tracker
    from rumi_slam_amd.vocabulary import ORBVocabulary
    trk = Tracker(1000, 1.2, 8, 20, 7, W, H, 4096)
    orc = O.OracleExtractor(1000, 1.2, 8, 20, 7)
    img0, sf, inv_sigma2, pts, last = _scene(n_keep=0.9)
    n0 = len(pts["obs"])
    rng = np.random.default_rng(5)
    pts["obs"] = np.where(rng.random(n0) < 0.1, 0, 2).astype(np.int32)
    pts["bad"] = (rng.random(n0) < 0.04).astype(np.uint8)
    voc_nodes = synthetic_vocabulary(21, 10, 3)
    # a vocabulary whose words are the scene's own descriptors makes true correspondences share words, as a trained one does
    parent, leaf, vdesc, weight = (a.copy() for a in voc_nodes)
    leaves = np.nonzero(leaf)[0]
    take = rng.choice(n0, len(leaves), replace=len(leaves) > n0)
    vdesc[leaves] = pts["desc"][take]
    g, o = ORBVocabulary(parent, leaf, vdesc, weight), O.OracleVocabulary(parent, leaf, vdesc, weight)
    q_gt, t_gt = _pose_gt(2)
    img = warp_homography(img0, _homography(q_gt, t_gt))
    mono, keys, desc = trk.extract(img)
    T_bad = np.array([0, 0, 0, 1, 9.0, 0, 0], np.float32)
    rm = _oracle_motion(keys, desc, sf, inv_sigma2, T_bad, last, pts)
    gm = trk.motion(K_TUM3, T_bad, last["keys"], last["mp"], last["outlier"], pts)
    assert rm["nmatches_motion"] < 20 and gm["nmatches_motion"] == rm["nmatches_motion"] and gm["th_motion"] == 30
    assert gm["ngood_motion"] == 0 and (gm["discarded"] == -1).all() and np.array_equal(gm["frame_mp"], rm["frame_mp"]) and np.array_equal(gm["Tcw_motion"], T_bad)
    # --- TrackReferenceKeyFrame: the key-frame is frame 0 (the last frame's features with their map points)
    levelsup = 2
    kf_keys, kf_desc, kf_mp = last["keys"], pts["desc"], last["mp"]       # frame-0 descriptors are the map points' descriptors in this scene
    (_, _), (kn, ko, ki) = o.transform(kf_desc, levelsup)
    wq, vq, nq = o.transform_features(desc, levelsup)
    (_, _), (fn, fo, fi) = o.transform(desc, levelsup)
    nm_ref, vp = O.search_by_bow(kf_keys, kf_desc, kf_mp, pts["bad"], (kn, ko, ki), keys, desc, (fn, fo, fi), 0.7, True)
    T_init = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)                   # mLastFrame.GetPose()
    got = trk.reference_keyframe(g, K_TUM3, T_init, FrameView(kf_keys, kf_desc, W, H, sf), FeatureVector.from_csr(kn, ko, ki), kf_mp, pts, levelsup, 0.7, True)
    assert np.array_equal(got["word_id"], wq) and np.array_equal(got["node_id"], nq) and np.array_equal(got["word_weight"], vq), "Frame::ComputeBoW per feature"
    (bi, bv), (an, ao, ai) = g.assemble(got["word_id"], got["word_weight"], got["node_id"])
    (bi2, bv2), _ = o.transform(desc, levelsup)
    assert np.array_equal(bi, bi2) and bv.tobytes() == bv2.tobytes() and np.array_equal(an, fn) and np.array_equal(ao, fo) and np.array_equal(ai, fi)
    assert nm_ref >= 15, f"the scene must give the BoW search enough matches ({nm_ref})"
    assert got["nmatches_motion"] == nm_ref
    idx = np.nonzero(vp >= 0)[0]
    ng, T1, out = O.pose_optimization(pts["pos"][vp[idx]], np.stack([keys["x"][idx], keys["y"][idx]], 1), inv_sigma2[keys["octave"][idx]], K_TUM3, T_init)
    exp_mp, exp_dis = vp.copy(), np.full(len(keys), -1, np.int32)
    exp_dis[idx[out != 0]] = vp[idx[out != 0]]
    exp_mp[idx[out != 0]] = -1
    inl = idx[out == 0]
    assert got["ngood_motion"] == ng and got["nmatches_map"] == int((pts["obs"][vp[inl]] > 0).sum())
    assert np.array_equal(got["frame_mp"], exp_mp) and np.array_equal(got["discarded"], exp_dis)
    _pose_close(got["Tcw_motion"], T1, "pose after TrackReferenceKeyFrame")
    # --- and TrackLocalMap on top of it, still the same resident frame
    seen = _seen_from(got["discarded"], n0)
    rl = _oracle_local(keys, desc, sf, inv_sigma2, got["Tcw_motion"], got["frame_mp"], seen, pts, 1.0)
    gl = trk.local(K_TUM3, got["Tcw_motion"], got["frame_mp"], pts, seen, 1.0)
    for k in ("n_to_match", "nmatches_local", "ngood_local", "matches_inliers"):
        assert gl[k] == rl[k], f"local after the reference key-frame: {k} {gl[k]} vs {rl[k]}"
    for k in ("frame_mp", "outlier", "in_view"):
        assert np.array_equal(gl[k], rl[k]), f"local after the reference key-frame: {k}"
    _pose_close(gl["Tcw"], rl["Tcw"], "pose after the local map")


def test_reference_keyframe_with_too_few_matches_and_stage_order_errors():
    from rumi_slam_amd.capi import RumiError
    from rumi_slam_amd.matcher import FeatureVector, FrameView
    from rumi_slam_amd.tracker import Tracker
    from rumi_slam_amd.vocabulary import ORBVocabulary
    trk = Tracker(1000, 1.2, 8, 20, 7, W, H, 4096)
    img0, sf, inv_sigma2, pts, last = _scene(n_keep=0.9)
    T = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    with pytest.raises(RumiError):
        trk.motion(K_TUM3, T, last["keys"], last["mp"], last["outlier"], pts)            # no frame is resident yet
    voc = synthetic_vocabulary(3, 8, 3)
    g, o = ORBVocabulary(*voc), O.OracleVocabulary(*voc)
    mono, keys, desc = trk.extract(synth_frame(777))                                  # an unrelated image: the BoW search finds (almost) nothing
    (_, _), (kn, ko, ki) = o.transform(pts["desc"], 2)
    (_, _), (fn, fo, fi) = o.transform(desc, 2)
    nm_ref, vp = O.search_by_bow(last["keys"], pts["desc"], last["mp"], pts["bad"], (kn, ko, ki), keys, desc, (fn, fo, fi), 0.7, True)
    got = trk.reference_keyframe(g, K_TUM3, T, FrameView(last["keys"], pts["desc"], W, H, sf), FeatureVector.from_csr(kn, ko, ki), last["mp"], pts, 2, 0.7, True)
    assert nm_ref < 15 and got["nmatches_motion"] == nm_ref and np.array_equal(got["frame_mp"], vp)
    assert got["ngood_motion"] == 0 and (got["discarded"] == -1).all() and np.array_equal(got["Tcw_motion"], T)


def test_reference_keyframe_on_orbvoc_geometry():
    """TrackReferenceKeyFrame with the reference's real vocabulary shape: k = 10, L = 6 (1 111 111 nodes, a synthetic tree: ORBvoc.txt is a missing
    blob) and levelsup = 4, as Frame::ComputeBoW calls it (Frame.cc:763-768): the FeatureVectors group by the 100 level-2 nodes.  Per-feature
    transform, the assembled vectors, the BoW matches and whatever follows (fewer than 15 matches: nothing; otherwise the optimised pose and the
    discard list) equal the oracle chain."""
    from rumi_slam_amd.matcher import FeatureVector, FrameView
    from rumi_slam_amd.tracker import Tracker
    from rumi_slam_amd.vocabulary import ORBVocabulary
    from voc_scene import synthetic_vocabulary_fast
    trk = Tracker(1000, 1.2, 8, 20, 7, W, H, 4096)
    img0, sf, inv_sigma2, pts, last = _scene(n_keep=0.9)
    n0 = len(pts["obs"])
    rng = np.random.default_rng(6)
    pts["obs"] = np.where(rng.random(n0) < 0.1, 0, 2).astype(np.int32)
    pts["bad"] = (rng.random(n0) < 0.04).astype(np.uint8)
    voc = synthetic_vocabulary_fast(21, 10, 6)
    g, o = ORBVocabulary(*voc), O.OracleVocabulary(*voc)
    q_gt, t_gt = _pose_gt(2)
    img = warp_homography(img0, _homography(q_gt, t_gt))
    mono, keys, desc = trk.extract(img)
    levelsup = 4
    kf_keys, kf_desc, kf_mp = last["keys"], pts["desc"], last["mp"]
    (_, _), (kn, ko, ki) = o.transform(kf_desc, levelsup)
    wq, vq, nq = o.transform_features(desc, levelsup)
    (bi2, bv2), (fn, fo, fi) = o.transform(desc, levelsup)
    assert 11 <= fn.min() and fn.max() <= 110, "level-2 node ids"
    nm_ref, vp = O.search_by_bow(kf_keys, kf_desc, kf_mp, pts["bad"], (kn, ko, ki), keys, desc, (fn, fo, fi), 0.7, True)
    T_init = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    got = trk.reference_keyframe(g, K_TUM3, T_init, FrameView(kf_keys, kf_desc, W, H, sf), FeatureVector.from_csr(kn, ko, ki), kf_mp, pts, levelsup, 0.7, True)
    assert np.array_equal(got["word_id"], wq) and np.array_equal(got["node_id"], nq) and np.array_equal(got["word_weight"], vq), "Frame::ComputeBoW per feature"
    (bi, bv), (an, ao, ai) = g.assemble(got["word_id"], got["word_weight"], got["node_id"])
    assert np.array_equal(bi, bi2) and bv.tobytes() == bv2.tobytes() and np.array_equal(an, fn) and np.array_equal(ao, fo) and np.array_equal(ai, fi)
    print(f"BoW matches on the k=10, L=6 tree at levelsup 4: {nm_ref}")
    assert got["nmatches_motion"] == nm_ref
    if nm_ref < 15:
        assert np.array_equal(got["frame_mp"], vp) and got["ngood_motion"] == 0 and (got["discarded"] == -1).all() and np.array_equal(got["Tcw_motion"], T_init)
        return
    idx = np.nonzero(vp >= 0)[0]
    ng, T1, out = O.pose_optimization(pts["pos"][vp[idx]], np.stack([keys["x"][idx], keys["y"][idx]], 1), inv_sigma2[keys["octave"][idx]], K_TUM3, T_init)
    exp_mp, exp_dis = vp.copy(), np.full(len(keys), -1, np.int32)
    exp_dis[idx[out != 0]] = vp[idx[out != 0]]
    exp_mp[idx[out != 0]] = -1
    assert got["ngood_motion"] == ng and np.array_equal(got["frame_mp"], exp_mp) and np.array_equal(got["discarded"], exp_dis)
    _pose_close(got["Tcw_motion"], T1, "pose after TrackReferenceKeyFrame")


def test_discarded_outliers_with_a_stale_in_view_flag_are_searched_at_their_old_projection():
    """Advisor finding of round 3 (low): the reference's "discard outliers" loop tests `i < mCurrentFrame.Nleft`; a monocular frame has Nleft = -1
    (Frame.cc:420), so the loop clears mbTrackInViewR and LEAVES mbTrackInView as the previous frame's SearchLocalPoints set it.  SearchLocalPoints
    then skips the point (mnLastFrameSeen == mnId) and SearchByProjection searches it with its old mTrackProjX / Y, level and viewing cosine
    (ORBmatcher.cc:46-60).  The device path now does the same when the caller hands the flags over (RumiTrackPoints.stale_in_view / stale_proj):
    fused call and step-wise calls against the oracle chain, on a frame where a few map points are wrong enough to be discarded."""
    from rumi_slam_amd.tracker import Tracker
    fused, steps = Tracker(1000, 1.2, 8, 20, 7, W, H, 4096), Tracker(1000, 1.2, 8, 20, 7, W, H, 4096)
    orc = O.OracleExtractor(1000, 1.2, 8, 20, 7)
    img0, sf, inv_sigma2, pts, last = _scene(n_keep=0.6)
    n0 = len(pts["obs"])
    rng = np.random.default_rng(11)
    wrong = rng.choice(np.nonzero(last["mp"] >= 0)[0], 40, replace=False)          # 40 points of the last frame sit 6-9 cm off: matched, then discarded
    pts["pos"][wrong, :2] += rng.choice([-1, 1], (40, 2)) * rng.uniform(0.06, 0.09, (40, 2)).astype(np.float32)
    # what the previous frame's SearchLocalPoints left in every MapPoint: in view, at the feature it was seen at in frame 0
    dist = np.linalg.norm(pts["pos"], axis=1).astype(np.float32)
    pts["stale_in_view"] = (rng.random(n0) < 0.7).astype(np.uint8)
    pts["stale_proj"] = np.stack([last["keys"]["x"], last["keys"]["y"], last["keys"]["octave"].astype(np.float32), np.ones(n0, np.float32), dist], 1).astype(np.float32)
    q_gt, t_gt = _pose_gt(2)
    img = warp_homography(img0, _homography(q_gt, t_gt))
    T0 = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    ref = _oracle_step(orc, img, sf, inv_sigma2, T0, last, pts, 3.0)
    n_stale = int((ref["in_view"] == 2).sum())
    assert n_stale >= 5, f"the scene must discard points that carry the stale flag ({n_stale})"
    got = fused.track(img, K_TUM3, T0, last["keys"], last["mp"], last["outlier"], pts, 15.0, 3.0)
    for k in ("nmatches_motion", "ngood_motion", "nmatches_map", "n_to_match", "nmatches_local", "ngood_local", "matches_inliers"):
        assert got[k] == ref[k], f"fused: {k} {got[k]} vs {ref[k]}"
    for k in ("frame_mp_motion", "in_view", "frame_mp", "outlier"):
        assert np.array_equal(got[k], ref[k]), f"fused: {k}"
    # without the flags the same call must differ (otherwise the path is untested): the stale points are then simply not searched
    plain = dict(pts); plain.pop("stale_in_view"); plain.pop("stale_proj")
    ref_plain = _oracle_step(orc, img, sf, inv_sigma2, T0, last, plain, 3.0)
    assert (ref_plain["in_view"] == 2).sum() == 0
    rematched = int((np.isin(ref["frame_mp"], np.nonzero(ref["in_view"] == 2)[0])).sum())
    print(f"stale in-view flags: {n_stale} discarded points searched at their old projection, {rematched} of them matched again; nmatches_local {ref['nmatches_local']} vs {ref_plain['nmatches_local']} without")
    # step-wise: motion -> (host discard loop) -> local with seen_in and the same flags
    steps.extract(img)
    gm = steps.motion(K_TUM3, T0, last["keys"], last["mp"], last["outlier"], pts)
    seen = _seen_from(gm["discarded"], n0)
    rl = _oracle_local(ref["keys"], ref["desc"], sf, inv_sigma2, gm["Tcw_motion"], gm["frame_mp"], seen, pts, 3.0)
    gl = steps.local(K_TUM3, gm["Tcw_motion"], gm["frame_mp"], pts, seen, 3.0)
    for k in ("n_to_match", "nmatches_local", "ngood_local", "matches_inliers"):
        assert gl[k] == rl[k], f"step-wise: {k} {gl[k]} vs {rl[k]}"
    for k in ("frame_mp", "outlier", "in_view"):
        assert np.array_equal(gl[k], rl[k]), f"step-wise: {k}"
    assert np.array_equal(gl["frame_mp"], got["frame_mp"]) and np.array_equal(gl["in_view"], got["in_view"])


EUROC_DIST = np.array([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0], np.float32)       # R/config/euroc_ori.yaml:23-31 (k1, k2, p1, p2; no k3)
EUROC_K = np.array([458.654, 457.296, 367.215, 248.375], np.float32)


@pytest.mark.parametrize("w,h", [(752, 480), (600, 350)])
def test_undistorted_keypoints_and_image_bounds(w, h):
    """Frame::UndistortKeyPoints / ComputeImageBounds (Frame.cc:770-826) on the resident frame, with euroc_ori.yaml's coefficients on the sensor's
    752 x 480 and on the 600 x 350 the reference resizes to (Camera.newWidth / newHeight: the intrinsics scale with the image, the coefficients
    stay): mvKeysUn and mnMinX .. mnMaxY bit-equal to the oracle's restatement of cv::undistortPoints (parity unpinned against OpenCV itself);
    mvKeys unchanged; switching the distortion off gives mvKeysUn = mvKeys and the image rectangle again."""
    from rumi_slam_amd.tracker import Tracker
    K = (EUROC_K * np.array([w / 752.0, h / 480.0, w / 752.0, h / 480.0])).astype(np.float32)
    trk = Tracker(1000, 1.2, 8, 20, 7, w, h, 1024)
    trk.set_distortion(K, EUROC_DIST)
    img = synth_frame(321, w=w, h=h)
    mono, keys, desc = trk.extract(img)
    okeys = O.OracleExtractor(1000, 1.2, 8, 20, 7).extract(img, (0, 1000))[1]
    assert keys.tobytes() == okeys.tobytes(), "mvKeys are the extractor's key-points"
    kun, bounds = trk.undistorted()
    ref = O.undistort_keys(keys, K, EUROC_DIST)
    assert kun.tobytes() == ref.tobytes(), "mvKeysUn"
    assert np.array_equal(bounds, O.image_bounds(w, h, K, EUROC_DIST)), "mnMinX .. mnMaxY"
    shift = np.hypot(kun["x"] - keys["x"], kun["y"] - keys["y"])
    assert shift.max() > 5 and bounds[0] < -5 and bounds[2] > w + 5, f"barrel distortion moves the corners outwards ({shift.max():.1f} px, bounds {bounds})"
    trk.set_distortion(K, None)
    trk.extract(img)
    kun2, b2 = trk.undistorted()
    assert kun2.tobytes() == keys.tobytes() and list(b2) == [0, 0, w, h]


def test_tracking_steps_with_lens_distortion():
    """TrackWithMotionModel and TrackLocalMap on a frame of a distorted camera: the grid is built from mvKeysUn over the undistorted bounds, the
    searches, the frustum test and PoseOptimization read mvKeysUn -- against the oracle chain given the same undistorted key-points and bounds."""
    from rumi_slam_amd.tracker import Tracker
    trk = Tracker(1000, 1.2, 8, 20, 7, W, H, 4096)
    dist = np.array([-0.2834, 0.0740, 0.0002, 0.00002, 0.0], np.float32)
    trk.set_distortion(K_TUM3, dist)
    img0, sf, inv_sigma2, pts, last = _scene(n_keep=0.6)
    # the map: frame 0's UNDISTORTED key-points back-projected onto the plane; the last frame holds frame 0's undistorted key-points
    from test_tracking_loop_gpu import PLANE_D
    ku0 = O.undistort_keys(last["keys"], K_TUM3, dist)
    fx, fy, cx, cy = K_TUM3.astype(np.float64)
    pos = np.stack([(ku0["x"] - cx) / fx * PLANE_D, (ku0["y"] - cy) / fy * PLANE_D, np.full(len(ku0), PLANE_D)], 1).astype(np.float32)
    d0 = np.linalg.norm(pos, axis=1).astype(np.float32)
    pts.update(pos=pos, normal=(pos / d0[:, None]).astype(np.float32), max_dist=(d0 * sf[ku0["octave"]]).astype(np.float32),
               min_dist=(d0 * sf[ku0["octave"]] / sf[7]).astype(np.float32))
    last = dict(last, keys=ku0)
    q_gt, t_gt = _pose_gt(1)
    img = warp_homography(img0, _homography(q_gt, t_gt))
    mono, keys, desc = trk.extract(img)
    kun, bounds = trk.undistorted()
    assert kun.tobytes() == O.undistort_keys(keys, K_TUM3, dist).tobytes() and np.array_equal(bounds, O.image_bounds(W, H, K_TUM3, dist))
    T0 = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    rm = _oracle_motion(kun, desc, sf, inv_sigma2, T0, last, pts, bounds=tuple(bounds))
    gm = trk.motion(K_TUM3, T0, last["keys"], last["mp"], last["outlier"], pts)
    assert rm["nmatches_motion"] >= 100, f"the scene must track ({rm['nmatches_motion']} matches)"
    for k in ("th_motion", "nmatches_motion", "ngood_motion", "nmatches_map"):
        assert gm[k] == rm[k], f"motion: {k} {gm[k]} vs {rm[k]}"
    assert np.array_equal(gm["frame_mp"], rm["frame_mp"]) and np.array_equal(gm["discarded"], rm["discarded"])
    _pose_close(gm["Tcw_motion"], rm["Tcw_motion"], "pose after the motion model")
    seen = _seen_from(gm["discarded"], len(pts["obs"]))
    rl = _oracle_local(kun, desc, sf, inv_sigma2, gm["Tcw_motion"], gm["frame_mp"], seen, pts, 3.0, bounds=tuple(bounds))
    gl = trk.local(K_TUM3, gm["Tcw_motion"], gm["frame_mp"], pts, seen, 3.0)
    for k in ("n_to_match", "nmatches_local", "ngood_local", "matches_inliers"):
        assert gl[k] == rl[k], f"local: {k} {gl[k]} vs {rl[k]}"
    for k in ("frame_mp", "outlier", "in_view"):
        assert np.array_equal(gl[k], rl[k]), f"local: {k}"
    _pose_close(gl["Tcw"], rl["Tcw"], "pose after the local map")
    # the fused call on the same inputs
    fused = Tracker(1000, 1.2, 8, 20, 7, W, H, 4096)
    fused.set_distortion(K_TUM3, dist)
    gf = fused.track(img, K_TUM3, T0, last["keys"], last["mp"], last["outlier"], pts, 15.0, 3.0)
    assert np.array_equal(gf["frame_mp"], gl["frame_mp"]) and np.array_equal(gf["outlier"], gl["outlier"]) and gf["matches_inliers"] == gl["matches_inliers"]
    print(f"lens distortion: motion {rm['nmatches_motion']} matches, local +{rl['nmatches_local']}, inliers {rl['matches_inliers']}; bounds {bounds}")
