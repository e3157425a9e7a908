#!/usr/bin/env python3
"""Measurement matrix of SURVEY.md §8(d) beyond the single bench.py line: batch sweep (single-stream .. 1024 frames), feature
count sweep (1000 / 2000 = TUM3.yaml / 5000 = mono-init extractor), the low-texture retry set, the windowed matchers M1-M3
through the host C ABI, brute force in pairs/s and Gpopc/s, and the all-cores CPU baseline of config 5.
Writes one JSON document (default gpurun_out/bench_matrix.json; the judged copy lives in profiles/)."""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def timed(fn, reps, sync):
    fn(); sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    sync()
    return (time.perf_counter() - t0) / reps


def median_call(fn, reps=50):
    """Median latency of a synchronous call (a mean is dominated by the interpreter's occasional gen-2 GC pause, ~50 ms with torch loaded)."""
    import numpy as np
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "bench_matrix.json"))
    ap.add_argument("--cpu-threads", type=int, default=min(16, os.cpu_count() or 1))
    args = ap.parse_args()
    import numpy as np
    import torch
    import oracle_lib as O
    from rumi_slam_amd.extractor import ORBextractor
    from rumi_slam_amd.matcher import FrameView, FeatureVector, ORBmatcher, bruteforce_batch
    from rumi_slam_amd.synth import synth_frame
    from scene import K_TUM3, TrackingScene
    dev = torch.device("cuda", 0)
    sync = torch.cuda.synchronize
    doc = {"device": torch.cuda.get_device_name(0), "host_cpus": os.cpu_count()}
    uniq = np.stack([synth_frame(1234 + i) for i in range(32)])

    def frames_for(B, src=uniq):
        t = torch.from_numpy(src).to(dev)
        return t.repeat((B + len(src) - 1) // len(src), 1, 1)[:B].contiguous()

    # ---- batch sweep, 1000 features ----
    rows = []
    for B in (1, 16, 64, 256, 1024):
        ext = ORBextractor(1000, 1.2, 8, 20, 7, max_batch=B)
        fr = frames_for(B)
        cap = 1000 + 96
        reps = max(3, min(50, 2048 // B))
        t_e = timed(lambda: ext.extract_batch(fr, (0, 1000), cap=cap), reps, sync)

        def both():
            kp, d, c = ext.extract_batch(fr, (0, 1000), cap=cap)
            bruteforce_batch(d, c, torch.roll(d, -1, 0), torch.roll(c, -1, 0))
        t_b = timed(both, reps, sync)
        rows.append({"frames_per_launch": B, "extract_fps": round(B / t_e, 1), "extract_match_fps": round(B / t_b, 1),
                     "ms_per_batch_extract": round(t_e * 1e3, 3)})
        ext.close()
    doc["batch_sweep_1000_features"] = rows

    # ---- single stream through the host API (PCIe both ways) ----
    ext = ORBextractor(1000, 1.2, 8, 20, 7)
    t1 = timed(lambda: ext(uniq[0]), 100, sync)
    doc["single_frame_host_api"] = {"ms_per_frame": round(t1 * 1e3, 3), "fps": round(1 / t1, 1), "note": "rumi_orb_extract: H2D image, kernels, D2H key-points + descriptors"}
    ext.close()

    # ---- feature-count sweep and the low-texture set at 256 frames per launch ----
    rows = []
    low = np.stack([synth_frame(3000 + i, n_rect=60, contrast=(8, 19)) for i in range(16)])
    sparse = np.stack([synth_frame(5000 + i, n_rect=40) for i in range(16)])
    for nf, src, tag in ((1000, uniq, "textured"), (2000, uniq, "textured"), (5000, uniq, "textured"), (1000, low, "low texture (minThFAST retry)"),
                         (1000, sparse, "sparse texture (40 rectangles: corner density of natural images)")):
        B = 256
        ext = ORBextractor(nf, 1.2, 8, 20, 7, max_batch=B)
        fr = frames_for(B, src)
        cap = nf + 96
        t_e = timed(lambda: ext.extract_batch(fr, (0, 1000), cap=cap), 5, sync)
        kp, d, c = ext.extract_batch(fr, (0, 1000), cap=cap)
        sync()
        rows.append({"nfeatures": nf, "input": tag, "frames_per_launch": B, "extract_fps": round(B / t_e, 1),
                     "mean_keypoints": round(float(c[:, 0].float().mean().item()), 1)})
        ext.close()
    doc["feature_sweep"] = rows

    # ---- brute force alone ----
    ext = ORBextractor(1000, 1.2, 8, 20, 7, max_batch=256)
    fr = frames_for(256)
    kp, d, c = ext.extract_batch(fr, (0, 1000), cap=1096)
    d2, c2 = torch.roll(d, -1, 0).contiguous(), torch.roll(c, -1, 0).contiguous()
    t_m = timed(lambda: bruteforce_batch(d, c, d2, c2), 20, sync)
    n = c[:, 0].double()
    pairs = float((n * torch.roll(n, -1, 0)).sum().item())
    doc["bruteforce"] = {"frame_pairs_per_s": round(256 / t_m, 1), "descriptor_pairs_per_s": round(pairs / t_m, 1),
                         "Gpopc64_per_s": round(pairs * 4 / t_m / 1e9, 2), "ms_per_256_pairs": round(t_m * 1e3, 3)}
    ext.close()

    # ---- windowed matchers through the host C ABI (uploads included) vs the oracle, one frame pair ----
    s = TrackingScene(0)
    m = ORBmatcher(0.8, True)
    F = FrameView(s.cur_keys, s.cur_desc, s.w, s.h, s.sf)
    mp = s.mappoint_view()
    fm0 = np.full(F.n, -1, np.int32)
    nosync = lambda: None
    cpu_time = lambda fn: median_call(fn, 20)
    g1 = median_call(lambda: m.SearchByProjection_MapPoints(F, mp, fm0, 3.0))
    c1 = cpu_time(lambda: O.search_by_projection_mappoints(s.cur_keys, s.cur_desc, s.w, s.h, s.sf, mp, fm0, 3.0, False, 0.0, 0.8))
    a2 = (s.Tcw7, K_TUM3, s.last_keys, s.last_mp, s.last_outlier, s.mp_pos, s.mp_desc, s.mp_obs, fm0)
    g2 = median_call(lambda: m.SearchByProjection_Frame(F, *a2, 15.0))
    c2_ = cpu_time(lambda: O.search_by_projection_frame(s.cur_keys, s.cur_desc, s.w, s.h, s.sf, *a2, 15.0, True))
    fv1, fv2 = s.feature_vectors()
    A, Bv = FeatureVector(fv1), FeatureVector(fv2)
    KF = FrameView(s.last_keys, s.last_desc, s.w, s.h, s.sf)
    bad = np.zeros(len(s.mp_obs), np.uint8)
    g3 = median_call(lambda: m.SearchByBoW(KF, A, s.last_mp, bad, F, Bv))
    c3 = cpu_time(lambda: O.search_by_bow(s.last_keys, s.last_desc, s.last_mp, bad, (A.node_ids, A.offsets, A.indices), s.cur_keys, s.cur_desc,
                                          (Bv.node_ids, Bv.offsets, Bv.indices), 0.8, True))
    # fused Tracking::SearchLocalPoints (frustum test + M1 in one call) against the two calls it replaces
    from rumi_slam_amd.matcher import isInFrustum
    from scene import quat_rotate
    gpts = s.point_geometry()
    _, Rm = quat_rotate(s.Tcw7[:4].astype(np.float64), np.zeros((1, 3)))
    R32 = Rm.astype(np.float32).ravel()
    log_sf = float(np.log(np.float32(1.2)))
    pts = dict(pos=s.mp_pos, normal=gpts["normal"], min_dist=gpts["min_dist"], max_dist=gpts["max_dist"], desc=s.mp_desc, obs=s.mp_obs,
               skip=np.zeros(len(s.mp_obs), np.uint8))

    def two_calls():
        f = isInFrustum(m, R32, s.Tcw7[4:], gpts["Ow"], K_TUM3, s.w, s.h, log_sf, 8, 0.5, pts)
        return m.SearchByProjection_MapPoints(F, dict(f, is_bad=pts["skip"], desc=s.mp_desc, obs=s.mp_obs), fm0, 3.0)

    def two_calls_cpu():
        f = O.is_in_frustum(R32, s.Tcw7[4:], gpts["Ow"], K_TUM3, s.w, s.h, log_sf, 8, 0.5, pts)
        return O.search_by_projection_mappoints(s.cur_keys, s.cur_desc, s.w, s.h, s.sf, dict(f, is_bad=pts["skip"], desc=s.mp_desc, obs=s.mp_obs), fm0, 3.0, False, 0.0, 0.8)
    g_two = median_call(two_calls)
    g_fused = median_call(lambda: m.SearchLocalPoints(F, R32, s.Tcw7[4:], gpts["Ow"], K_TUM3, log_sf, 8, pts, fm0, 3.0))
    c_two = cpu_time(two_calls_cpu)
    doc["search_local_points_host_api"] = {"map_points": int(len(s.mp_obs)), "gpu_us_fused": round(g_fused * 1e6, 1), "gpu_us_two_calls": round(g_two * 1e6, 1),
                                           "cpu_oracle_us": round(c_two * 1e6, 1),
                                           "note": "Tracking::SearchLocalPoints: Frame::isInFrustum for every local point + SearchByProjection(F, points); fused = rumi_search_local_points"}
    doc["windowed_matchers_host_api"] = {
        "note": "median latency of one call per frame pair, host arrays in/out (one pinned upload, kernels, one read-back); oracle = scalar CPU restatement",
        "M1_SearchByProjection_mappoints": {"queries": int(len(mp["obs"])), "gpu_us": round(g1 * 1e6, 1), "cpu_oracle_us": round(c1 * 1e6, 1)},
        "M2_SearchByProjection_frame": {"queries": int(len(s.last_keys)), "gpu_us": round(g2 * 1e6, 1), "cpu_oracle_us": round(c2_ * 1e6, 1)},
        "M3_SearchByBoW": {"queries": int(len(s.last_keys)), "gpu_us": round(g3 * 1e6, 1), "cpu_oracle_us": round(c3 * 1e6, 1)}}

    # ---- one Tracking-thread frame through the host API: extract -> SearchByProjection(Cur, Last) -> PoseOptimization ->
    #      SearchByProjection(F, local points) -> PoseOptimization (TrackWithMotionModel + TrackLocalMap, Tracking.cc:2434-2560)
    from rumi_slam_amd.optimizer import Optimizer
    ext1 = ORBextractor(1000, 1.2, 8, 20, 7)
    orc1 = O.OracleExtractor(1000, 1.2, 8, 20, 7)
    opt = Optimizer()
    inv_s2 = ext1.GetInverseScaleSigmaSquares()
    img = synth_frame(9000)                       # the frame TrackingScene(0) calls "last"; its warped successor is s.cur_*
    from rumi_slam_amd.synth import warp_frame
    img_cur, _ = warp_frame(img, 777)

    def track_gpu():
        _, keys, desc = ext1(img_cur)
        Fc = FrameView(keys, desc, s.w, s.h, s.sf)
        nm, cur_mp = m.SearchByProjection_Frame(Fc, *a2[:-1], np.full(Fc.n, -1, np.int32), 15.0)
        idx = np.nonzero(cur_mp >= 0)[0]
        ng, T, out = opt.PoseOptimization(s.mp_pos[cur_mp[idx]], np.stack([keys["x"][idx], keys["y"][idx]], 1), inv_s2[keys["octave"][idx]], K_TUM3, s.Tcw7)
        n2, fm = m.SearchByProjection_MapPoints(Fc, mp, cur_mp, 3.0)
        idx = np.nonzero((fm >= 0) & (fm < len(s.mp_pos)))[0]
        opt.PoseOptimization(s.mp_pos[fm[idx]], np.stack([keys["x"][idx], keys["y"][idx]], 1), inv_s2[keys["octave"][idx]], K_TUM3, T)
        return ng

    def track_cpu():
        _, keys, desc = orc1.extract(img_cur)
        nm, cur_mp = O.search_by_projection_frame(keys, desc, s.w, s.h, s.sf, *a2[:-1], np.full(len(keys), -1, np.int32), 15.0, True)
        idx = np.nonzero(cur_mp >= 0)[0]
        ng, T, out = O.pose_optimization(s.mp_pos[cur_mp[idx]], np.stack([keys["x"][idx], keys["y"][idx]], 1), inv_s2[keys["octave"][idx]], K_TUM3, s.Tcw7)
        n2, fm = O.search_by_projection_mappoints(keys, desc, s.w, s.h, s.sf, mp, cur_mp, 3.0, False, 0.0, 0.8)
        idx = np.nonzero((fm >= 0) & (fm < len(s.mp_pos)))[0]
        O.pose_optimization(s.mp_pos[fm[idx]], np.stack([keys["x"][idx], keys["y"][idx]], 1), inv_s2[keys["octave"][idx]], K_TUM3, T)
        return ng

    assert track_gpu() == track_cpu()
    tg, tc = median_call(track_gpu, 30), median_call(track_cpu, 5)
    doc["tracking_frame_host_api"] = {"sequence": "ORBextractor() -> SearchByProjection(Cur,Last) -> PoseOptimization -> SearchByProjection(F, local points) -> PoseOptimization",
                                      "gpu_ms": round(tg * 1e3, 3), "cpu_oracle_ms": round(tc * 1e3, 2), "gpu_fps": round(1 / tg, 1), "cpu_fps": round(1 / tc, 1),
                                      "note": "one 640x480 frame, host arrays in and out at every call (PCIe included), python wrapper overhead included on both sides"}

    # ---- OptimizeSim3 (loop / merge verification) and OptimizeCloudSim3 (sub-map alignment) through the host ABI vs the oracle ----
    from sim3_scene import sim3_pair_problem, sim3_cloud_problem
    b = sim3_pair_problem(seed=1, n=400)
    ap_ = (b["S0"], b["P1c"], b["P2c"], b["obs1"], b["obs2"], b["w1"], b["w2"], b["K"], b["K"], 10.0, False, True)
    c = sim3_cloud_problem(seed=11)
    ac_ = (c["S0"], c["P1c"], c["P2c"], c["obs1"], c["obs2"], c["w1"], c["w2"], c["K"], c["K"], 10.0, True, False, c["pair_of"], c["S_c1w"], c["S_c2w"],
           c["skip12"], c["skip21"])
    doc["optimize_sim3_host_api"] = {
        "note": "median latency of one call, host arrays in/out; one 256-thread workgroup runs both optimize() calls with g2o's numeric Jacobians",
        "OptimizeSim3": {"correspondences": 400, "gpu_us": round(median_call(lambda: opt.OptimizeSim3(*ap_), 20) * 1e6, 1),
                         "cpu_oracle_us": round(median_call(lambda: O.optimize_sim3(*ap_), 5) * 1e6, 1)},
        "OptimizeCloudSim3": {"correspondences": int(len(c["pair_of"])), "key_frame_pairs": int(len(c["S_c1w"])),
                              "gpu_us": round(median_call(lambda: opt.OptimizeSim3(*ac_), 20) * 1e6, 1),
                              "cpu_oracle_us": round(median_call(lambda: O.optimize_sim3(*ac_), 5) * 1e6, 1)}}

    # ---- Sim3Solver::iterate of the sub-map merge: a block of RANSAC iterations per launch (each one: Horn + CheckInliers + ComputeInliersNum) ----
    from sim3_scene import sim3_ransac_problem
    rp = sim3_ransac_problem(0, n_pairs=10, per_pair=150, n_solver=200)
    ra_ = (rp["X1"], rp["X2"], rp["sigma2_1"], rp["sigma2_2"], rp["K"], rp["K"])
    doc["sim3_ransac_host_api"] = {"note": "Sim3Solver::iterate (rumination overload): every iteration scores its hypothesis over all key-frame pairs; one workgroup "
                                           "per iteration, host arrays in/out; oracle = scalar CPU restatement of the same iterations",
                                   "correspondences": int(len(rp["X1"])), "scored_matches": int(rp["score"]["pair_start"][-1]),
                                   "key_frame_pairs": int(len(rp["score"]["pair_start"]) - 1)}
    for H in (20, 300):
        tri = O.sim3_draw_triples(0, len(rp["X1"]), H)
        doc["sim3_ransac_host_api"][f"iterations_{H}"] = {
            "gpu_us": round(median_call(lambda: opt.Sim3Ransac(*ra_, tri, score=rp["score"]), 20) * 1e6, 1),
            "cpu_oracle_us": round(median_call(lambda: O.sim3_ransac(*ra_, tri, score=rp["score"]), 3) * 1e6, 1)}

    # ---- CPU oracle, all cores (config 5: one frame per thread) ----
    T = args.cpu_threads
    orcs = [O.OracleExtractor(1000, 1.2, 8, 20, 7) for _ in range(T)]
    def work(k):
        for i in range(6):
            orcs[k].extract(uniq[(k * 6 + i) % len(uniq)], (0, 1000))
    with ThreadPoolExecutor(T) as ex:
        t0 = time.perf_counter(); list(ex.map(work, range(T))); dt = time.perf_counter() - t0
    t0 = time.perf_counter(); work(0); dt1 = time.perf_counter() - t0
    doc["cpu_oracle_extract"] = {"threads": T, "fps_all_threads": round(T * 6 / dt, 1), "fps_one_thread": round(6 / dt1, 1), "flags": "g++ -O2 -ffp-contract=off"}

    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(doc, open(args.out, "w"), indent=1)
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
