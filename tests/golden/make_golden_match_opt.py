#!/usr/bin/env python3
"""Generates the committed golden vectors of the matchers and optimisers from the CPU oracle (oracle/), on the seeded synthetic
scenes of tests/scene.py, tests/ba_scene.py and tests/sim3_scene.py.

As for the extractor (make_golden.py): the reference holds no golden vectors / tests for this path and cannot be built here, so these
pin the ORACLE against regressions and give the GPU tests fixtures that are independent of the oracle library at run time.  Parity
against the reference binary is unpinned.  Every fixture stores a SHA-256 of its inputs, so a drift of the scene
generators is reported as such.

    python tests/golden/make_golden_match_opt.py      # rewrites tests/golden/{match,opt}_*.npz
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def match_case(seed):
    """M1 / M2 / M3 / SearchForInitialization on TrackingScene(seed): the frame's map-point vectors after each search."""
    import oracle_lib as O
    from scene import K_TUM3, TrackingScene
    s = TrackingScene(seed)
    mp = s.mappoint_view()
    fm0 = np.full(len(s.cur_keys), -1, np.int32)
    n1, fm1 = O.search_by_projection_mappoints(s.cur_keys, s.cur_desc, s.w, s.h, s.sf, mp, fm0, 3.0, False, 0.0, 0.8)
    a2 = (s.Tcw7, K_TUM3, s.last_keys, s.last_mp, s.last_outlier, s.mp_pos, s.mp_desc, s.mp_obs, fm0)
    n2, fm2 = O.search_by_projection_frame(s.cur_keys, s.cur_desc, s.w, s.h, s.sf, *a2, 15.0, True)
    fv1, fv2 = s.feature_vectors()
    bad = np.zeros(len(s.mp_obs), np.uint8)
    f = lambda fv: (np.array(sorted(fv), np.uint32), np.cumsum([0] + [len(fv[k]) for k in sorted(fv)]).astype(np.int32),
                    np.concatenate([np.asarray(fv[k], np.uint32) for k in sorted(fv)]))
    n3, fm3 = O.search_by_bow(s.last_keys, s.last_desc, s.last_mp, bad, f(fv1), s.cur_keys, s.cur_desc, f(fv2), 0.8, True)
    pm0 = np.stack([s.last_keys["x"], s.last_keys["y"]], 1).astype(np.float32)
    n4, m12, pm = O.search_for_initialization(s.last_keys, s.last_desc, s.cur_keys, s.cur_desc, s.w, s.h, pm0, 100, 0.9, True)
    return dict(seed=np.int32(seed), inputs_sha256=sha(s.cur_keys.view(np.uint8), s.cur_desc, s.last_keys.view(np.uint8), s.last_desc, s.mp_pos, s.Tcw7),
                m1_n=np.int32(n1), m1_frame_mp=fm1, m2_n=np.int32(n2), m2_cur_mp=fm2, m3_n=np.int32(n3), m3_matches=fm3,
                init_n=np.int32(n4), init_matches12=m12)


def opt_case():
    import oracle_lib as O
    from ba_scene import ba_problem, pose_problem
    from sim3_scene import sim3_cloud_problem, sim3_pair_problem
    p = pose_problem(41, 300, 0.1)
    ng, T, out = O.pose_optimization(p["Xw"], p["obs"], p["inv_sigma2"], p["K"], p["T0"])
    b = ba_problem(seed=42, n_opt=6, n_fixed=2, n_points=400)
    a = (b["kf_pose"], b["kf_fixed"], b["mp_pos"], b["e_mp"], b["e_kf"], b["e_obs"], b["e_w"], b["K"])
    its, kp, mp, er = O.local_ba(*a)
    gits, gkp, gmp = O.bundle_adjustment(*a, 10, True)
    s = sim3_pair_problem(seed=43, n=150)
    nin, nbad, early, S, st = O.optimize_sim3(s["S0"], s["P1c"], s["P2c"], s["obs1"], s["obs2"], s["w1"], s["w2"], s["K"], s["K"], 10.0, False, True)
    c = sim3_cloud_problem(seed=44, n_pairs=4, per_pair=80)
    cn, cb, ce, cS, cst = O.optimize_sim3(c["S0"], c["P1c"], c["P2c"], c["obs1"], c["obs2"], c["w1"], c["w2"], c["K"], c["K"], 10.0, True, False, c["pair_of"],
                                          c["S_c1w"], c["S_c2w"], c["skip12"], c["skip21"])
    return dict(pose_inputs_sha256=sha(p["Xw"], p["obs"], p["inv_sigma2"], p["T0"]), pose_n_good=np.int32(ng), pose_T=T, pose_outlier=out,
                ba_inputs_sha256=sha(*a), lba_iterations=np.int32(its), lba_kf=kp, lba_mp=mp, lba_erase=er, gba_iterations=np.int32(gits), gba_kf=gkp, gba_mp=gmp,
                sim3_inputs_sha256=sha(s["S0"], s["P1c"], s["P2c"], s["obs1"], s["obs2"]), sim3_counts=np.array([nin, nbad, early], np.int32), sim3_S=S, sim3_status=st,
                cloud_inputs_sha256=sha(c["S0"], c["P1c"], c["P2c"], c["obs1"], c["obs2"], c["S_c1w"], c["S_c2w"]), cloud_counts=np.array([cn, cb, ce], np.int32),
                cloud_S=cS, cloud_status=cst)


if __name__ == "__main__":
    for seed in (0, 3):
        out = match_case(seed)
        np.savez_compressed(os.path.join(HERE, f"match_scene{seed}.npz"), **out)
        print("match scene", seed, int(out["m1_n"]), int(out["m2_n"]), int(out["m3_n"]), int(out["init_n"]))
    out = opt_case()
    np.savez_compressed(os.path.join(HERE, "opt_problems.npz"), **out)
    print("opt", int(out["pose_n_good"]), int(out["lba_iterations"]), int(out["gba_iterations"]), out["sim3_counts"], out["cloud_counts"])
