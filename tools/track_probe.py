"""Latency of one Tracking-thread frame: the fused device-resident entry (rumi_track_frame) against the same five stages through the separate
host-array entries (extract -> SearchByProjection(Cur, Last) -> PoseOptimization -> SearchLocalPoints -> PoseOptimization)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from rumi_slam_amd.extractor import ORBextractor
from rumi_slam_amd.matcher import FrameView, ORBmatcher
from rumi_slam_amd.optimizer import Optimizer
from rumi_slam_amd.synth import synth_frame, warp_homography
from rumi_slam_amd.tracker import Tracker
from scene import K_TUM3
from test_tracking_loop_gpu import PLANE_D, _homography, _pose_gt
from test_track_frame_gpu import _pose_matrices

def measure(reps=60):
    W, H = 640, 480
    ext, m9, m8, opt = ORBextractor(1000, 1.2, 8, 20, 7), ORBmatcher(0.9, True), ORBmatcher(0.8, True), Optimizer()
    trk = Tracker(1000, 1.2, 8, 20, 7, W, H, 4096)
    sf, inv_sigma2 = ext.GetScaleFactors(), ext.GetInverseScaleSigmaSquares()
    img0 = synth_frame(4242)
    fx, fy, cx, cy = K_TUM3.astype(np.float64)
    _, keys0, desc0 = ext(img0)
    n0 = len(keys0)
    pos = np.stack([(keys0["x"] - cx) / fx * PLANE_D, (keys0["y"] - cy) / fy * PLANE_D, np.full(n0, PLANE_D)], 1).astype(np.float32)
    dist0 = np.linalg.norm(pos, axis=1).astype(np.float32)
    lvl = keys0["octave"]
    pts = dict(pos=pos, normal=(pos / dist0[:, None]).astype(np.float32), max_dist=(dist0 * sf[lvl]).astype(np.float32),
               min_dist=(dist0 * sf[lvl] / sf[7]).astype(np.float32), desc=desc0.copy(), obs=np.ones(n0, np.int32), bad=np.zeros(n0, np.uint8),
               local=np.ones(n0, np.uint8))
    rng = np.random.default_rng(7)
    known = rng.random(n0) < 0.5
    last = dict(keys=keys0, mp=np.where(known, np.arange(n0), -1).astype(np.int32), outlier=np.zeros(n0, np.uint8))
    T = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    img = warp_homography(img0, _homography(*_pose_gt(1)))
    log_sf = float(np.log(np.float32(1.2)))


    def separate():
        mono, keys, desc = ext(img)
        F = FrameView(keys, desc, W, H, sf)
        nm, cur = m9.SearchByProjection_Frame(F, T, K_TUM3, last["keys"], last["mp"], last["outlier"], pts["pos"], pts["desc"], pts["obs"], np.full(F.n, -1, np.int32), 15.0)
        idx = np.nonzero(cur >= 0)[0]
        ng, T1, out = opt.PoseOptimization(pts["pos"][cur[idx]], np.stack([keys["x"][idx], keys["y"][idx]], 1), inv_sigma2[keys["octave"][idx]], K_TUM3, T)
        seen = np.zeros(n0, np.uint8); seen[cur[idx]] = 1
        cur[idx[out != 0]] = -1
        R, t, Ow = _pose_matrices(T1)
        nto, nml, cur2, _ = m8.SearchLocalPoints(F, R, t, Ow, K_TUM3, log_sf, 8, dict(pts, skip=seen), cur, 1.0)
        idx2 = np.nonzero(cur2 >= 0)[0]
        ng2, T2, out2 = opt.PoseOptimization(pts["pos"][cur2[idx2]], np.stack([keys["x"][idx2], keys["y"][idx2]], 1), inv_sigma2[keys["octave"][idx2]], K_TUM3, T1)
        return ng2, T2


    def fused():
        r = trk.track(img, K_TUM3, T, last["keys"], last["mp"], last["outlier"], pts, 15.0, 1.0)
        return r["ngood_local"], r["Tcw"]


    pinned = trk.image_buffer(W, H)                                # the frame captured straight into the tracker's pinned staging memory
    pinned[:] = img
    def fused_pinned():
        r = trk.track(pinned, K_TUM3, T, last["keys"], last["mp"], last["outlier"], pts, 15.0, 1.0)
        return r["ngood_local"], r["Tcw"]


    def c_call_ms(reps):                                           # the C entry alone, without the Python mirror's array allocations and copies
        real, spent = trk._lib.rumi_track_frame, []
        def timed(*a):
            t0 = time.perf_counter(); rc = real(*a); spent.append(time.perf_counter() - t0); return rc
        trk._lib.rumi_track_frame = timed
        try:
            for _ in range(reps + 5): fused_pinned()
        finally:
            trk._lib.rumi_track_frame = real
        return float(np.median(spent[5:])) * 1e3


    def med(f, reps):
        for _ in range(5): f()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
        return float(np.median(ts)) * 1e3


    # ---- the step-wise entries on one resident frame: extract | motion | local, and extract | reference key-frame (BoW) | local
    from rumi_slam_amd.matcher import FeatureVector
    from rumi_slam_amd.vocabulary import ORBVocabulary
    from voc_scene import synthetic_vocabulary
    parent, leaf, vdesc, weight = (x.copy() for x in synthetic_vocabulary(21, 10, 3))
    leaves = np.nonzero(leaf)[0]
    vdesc[leaves] = desc0[rng.choice(n0, len(leaves), replace=len(leaves) > n0)]
    voc = ORBVocabulary(parent, leaf, vdesc, weight)
    (_, _), (kn, ko, ki) = voc.transform(desc0, 2)
    kfv, kview = FeatureVector.from_csr(kn, ko, ki), FrameView(keys0, desc0, W, H, sf)
    kf_mp = np.arange(n0, dtype=np.int32)
    stage = {}
    # the reference's real vocabulary geometry (ORBvoc.txt: k = 10, L = 6, 1.1 M nodes; Frame::ComputeBoW asks for levelsup = 4): a synthetic tree of that shape
    from voc_scene import synthetic_vocabulary_fast
    voc_big = ORBVocabulary(*synthetic_vocabulary_fast(21, 10, 6))
    (_, _), (bn, bo, bi_) = voc_big.transform(desc0, 4)
    kfv_big = FeatureVector.from_csr(bn, bo, bi_)

    def steps_motion():
        t0 = time.perf_counter(); trk.extract(img); t1 = time.perf_counter()
        m = trk.motion(K_TUM3, T, last["keys"], last["mp"], last["outlier"], pts); t2 = time.perf_counter()
        seen = np.zeros(n0, np.uint8); seen[m["discarded"][m["discarded"] >= 0]] = 1
        l = trk.local(K_TUM3, m["Tcw_motion"], m["frame_mp"], pts, seen, 1.0); t3 = time.perf_counter()
        stage.setdefault("extract", []).append(t1 - t0); stage.setdefault("motion", []).append(t2 - t1); stage.setdefault("local", []).append(t3 - t2)
        return l["ngood_local"], l["Tcw"]

    def steps_refkf():
        trk.extract(img); t1 = time.perf_counter()
        m = trk.reference_keyframe(voc_big, K_TUM3, T, kview, kfv_big, kf_mp, pts, 4, 0.7, True); t2 = time.perf_counter()
        stage.setdefault("reference_keyframe", []).append(t2 - t1)
        stage.setdefault("reference_keyframe_c_call", []).append(trk.last_call_s)
        return m["ngood_motion"], m["Tcw_motion"]

    def steps_refkf_small():
        trk.extract(img); t1 = time.perf_counter()
        m = trk.reference_keyframe(voc, K_TUM3, T, kview, kfv, kf_mp, pts, 2, 0.7, True); t2 = time.perf_counter()
        stage.setdefault("reference_keyframe_small_tree_k10_L3", []).append(t2 - t1)
        return m["ngood_motion"], m["Tcw_motion"]

    a, b = separate(), fused()
    assert a[0] == b[0] and np.allclose(a[1], b[1], atol=1e-5), (a, b)
    bp = fused_pinned()
    assert bp[0] == b[0] and np.array_equal(bp[1], b[1]), (bp, b)
    c = steps_motion()
    assert c[0] == b[0] and np.allclose(c[1], b[1], atol=1e-5), (c, b)
    med(steps_refkf_small, reps)
    step_ms, ref_ms = med(steps_motion, reps), med(steps_refkf, reps)       # (the big tree last: a kernel trace of this script ends with its calls)
    per = {k: round(float(np.median(v[5:])) * 1e3, 3) for k, v in stage.items()}
    return dict(step_wise_ms=round(step_ms, 3), extract_plus_reference_keyframe_ms=round(ref_ms, 3), stage_ms=per,
                reference_keyframe_vocabulary="synthetic tree of ORBvoc.txt's geometry: k = 10, L = 6, 1 111 111 nodes, levelsup = 4 (the k = 10, L = 3 tree of round 3 beside it)", workload="one Tracking-thread frame: extract (640x480, 1000 features) -> SearchByProjection(Cur, Last) -> PoseOptimization -> SearchLocalPoints (%d map points) -> PoseOptimization; host image in, host results out" % n0,
                inliers=int(b[0]), separate_entries_ms=round(med(separate, reps), 3), rumi_track_frame_ms=round(med(fused, reps), 3),
                rumi_track_frame_pinned_ms=round(med(fused_pinned, reps), 3), c_call_pinned_ms=round(c_call_ms(reps), 3))


if __name__ == "__main__":
    r = measure()
    print("%s: separate entries %.3f ms, rumi_track_frame %.3f ms (inliers %d); the frame captured into rumi_track_image_buffer's pinned memory: %.3f ms (the C call alone, without the Python mirror's allocations: %.3f ms)" %
          (r["workload"], r["separate_entries_ms"], r["rumi_track_frame_ms"], r["inliers"], r["rumi_track_frame_pinned_ms"], r["c_call_pinned_ms"]))
    print("step-wise entries on the resident frame: extract + motion + local %.3f ms; extract + reference key-frame %.3f ms; per call (ms): %s" %
          (r["step_wise_ms"], r["extract_plus_reference_keyframe_ms"], r["stage_ms"]))
