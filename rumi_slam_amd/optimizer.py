"""Host-side mirror of the two hot members of the all-static ``ORB_SLAM3::Optimizer``
(R/include/cloud_edge_slam_lib/Optimizer.h:53,55) over the C ABI in include/rumi_opt.h."""
import ctypes as C

import numpy as np

from . import capi


class RumiSim3ScoreSet(C.Structure):
    """include/rumi_opt.h RumiSim3ScoreSet"""
    _fields_ = [("n_pairs", C.c_int32)] + [(k, C.c_void_p) for k in ("pair_start", "pair_denominator", "S_c1w1", "S_c2w2", "S_kf1w", "S_kf2w", "K4_1", "K4_2",
                                                                   "X1", "X2", "kp1", "kp2", "sigma2_1", "sigma2_2", "edge1", "edge2")]


OPT_SYMBOLS = ["rumi_opt_create", "rumi_opt_destroy", "rumi_pose_optimization", "rumi_pose_optimization_batch", "rumi_local_ba", "rumi_merge_ba", "rumi_bundle_adjustment", "rumi_sim3_inliers",
               "rumi_optimize_sim3", "rumi_sim3_ransac", "rumi_opt_stage_ms", "rumi_opt_set_profiling", "rumi_opt_kernel_ms"]


def _lib():
    L = capi.lib()
    if getattr(L, "_opt_ready", False):
        return L
    vp, i32 = C.c_void_p, C.c_int32
    L.rumi_opt_create.argtypes = [i32, i32, i32, i32, i32, i32, C.POINTER(vp)]
    L.rumi_opt_destroy.argtypes = [vp]
    L.rumi_opt_destroy.restype = None
    L.rumi_pose_optimization.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp, C.POINTER(i32)]
    L.rumi_pose_optimization_batch.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.rumi_local_ba.argtypes = [vp, i32, vp, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.rumi_local_ba_batch.argtypes = [vp, i32, vp, i32]
    L.rumi_bundle_adjustment.argtypes = [vp, i32, vp, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, vp]
    L.rumi_sim3_inliers.argtypes = [vp, i32] + [vp] * 17
    L.rumi_merge_ba.argtypes = [vp, i32, vp, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.rumi_opt_stage_ms.argtypes = [vp, vp]
    L.rumi_opt_set_profiling.argtypes = [vp, i32]
    L.rumi_opt_kernel_ms.argtypes = [vp, vp]
    L.rumi_sim3_ransac.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    L.rumi_optimize_sim3.argtypes = [vp, i32, vp, i32] + [vp] * 12 + [C.c_float, i32, i32, vp, vp, vp]
    L._opt_ready = True
    return L


class RumiBaWindow(C.Structure):
    _fields_ = [("n_kf", C.c_int32), ("kf_pose7", C.c_void_p), ("kf_fixed", C.c_void_p), ("n_mp", C.c_int32), ("mp_pos3", C.c_void_p),
                ("n_edges", C.c_int32), ("e_mp", C.c_void_p), ("e_kf", C.c_void_p), ("e_obs", C.c_void_p), ("e_inv_sigma2", C.c_void_p),
                ("K4", C.c_void_p), ("stop_flag", C.c_void_p), ("erase_out", C.c_void_p), ("stats", C.c_int32 * 4), ("status", C.c_int32)]


class Optimizer:
    def __init__(self, max_pose_edges=1 << 18, max_pose_batch=1024, max_kf=64, max_mp=16384, max_edges=1 << 18, device=-1):
        self._lib = _lib()
        self._h = C.c_void_p()
        capi.check(self._lib.rumi_opt_create(max_pose_edges, max_pose_batch, max_kf, max_mp, max_edges, device, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.rumi_opt_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def PoseOptimization(self, Xw, obs, inv_sigma2, K4, Tcw7):
        """Returns (nInitialCorrespondences - nBad, Tcw7, outlier[n])."""
        Xw = np.ascontiguousarray(Xw, np.float32); obs = np.ascontiguousarray(obs, np.float32)
        w = np.ascontiguousarray(inv_sigma2, np.float32); K4 = np.ascontiguousarray(K4, np.float32)
        T = np.ascontiguousarray(Tcw7, np.float32).copy()
        out = np.zeros(max(len(w), 1), np.uint8)
        ng = C.c_int32()
        capi.check(self._lib.rumi_pose_optimization(self._h, capi.ptr(Xw), capi.ptr(obs), capi.ptr(w), len(w), capi.ptr(K4), capi.ptr(T),
                                                    capi.ptr(out), C.byref(ng)))
        return ng.value, T, out[:len(w)]

    def PoseOptimizationBatch(self, start, Xw, obs, inv_sigma2, K4, Tcw7):
        start = np.ascontiguousarray(start, np.int32)
        Xw = np.ascontiguousarray(Xw, np.float32); obs = np.ascontiguousarray(obs, np.float32)
        w = np.ascontiguousarray(inv_sigma2, np.float32); K4 = np.ascontiguousarray(K4, np.float32)
        T = np.ascontiguousarray(Tcw7, np.float32).copy()
        B = len(start) - 1
        out = np.zeros(max(len(w), 1), np.uint8)
        ng = np.zeros(B, np.int32)
        capi.check(self._lib.rumi_pose_optimization_batch(self._h, B, capi.ptr(start), capi.ptr(Xw), capi.ptr(obs), capi.ptr(w), capi.ptr(K4),
                                                          capi.ptr(T), capi.ptr(out), capi.ptr(ng)))
        return ng, T, out[:len(w)]

    def LocalBundleAdjustment(self, kf_pose, kf_fixed, mp_pos, e_mp, e_kf, e_obs, e_inv_sigma2, K4, stop_flag=None, merge=False):
        """Returns (stats[4], kf_pose, mp_pos, erase[nE])."""
        kp = np.ascontiguousarray(kf_pose, np.float32).copy(); kfix = np.ascontiguousarray(kf_fixed, np.uint8)
        mp = np.ascontiguousarray(mp_pos, np.float32).copy(); em = np.ascontiguousarray(e_mp, np.int32)
        ek = np.ascontiguousarray(e_kf, np.int32); eo = np.ascontiguousarray(e_obs, np.float32)
        ew = np.ascontiguousarray(e_inv_sigma2, np.float32); K4 = np.ascontiguousarray(K4, np.float32)
        erase = np.zeros(max(len(em), 1), np.uint8)
        stats = np.zeros(4, np.int32)
        sp = capi.ptr(stop_flag) if stop_flag is not None else None
        fn = self._lib.rumi_merge_ba if merge else self._lib.rumi_local_ba
        import time
        t0 = time.perf_counter()
        rc = fn(self._h, len(kfix), capi.ptr(kp), capi.ptr(kfix), len(mp), capi.ptr(mp), len(em), capi.ptr(em),
                capi.ptr(ek), capi.ptr(eo), capi.ptr(ew), capi.ptr(K4), sp, capi.ptr(erase), capi.ptr(stats))
        self.last_call_s = time.perf_counter() - t0
        capi.check(rc)
        return stats, kp, mp, erase[:len(em)]

    def LocalBundleAdjustmentBatch(self, windows, n_workers=4):
        """rumi_local_ba_batch: `windows` = list of (kf_pose, kf_fixed, mp_pos, e_mp, e_kf, e_obs, e_inv_sigma2, K4) tuples; returns one
        (stats[4], kf_pose, mp_pos, erase[nE]) per window, as LocalBundleAdjustment does."""
        arr = (RumiBaWindow * len(windows))()
        keep = []
        for W, w in zip(arr, windows):
            kp = np.ascontiguousarray(w[0], np.float32).copy(); kfix = np.ascontiguousarray(w[1], np.uint8)
            mp = np.ascontiguousarray(w[2], np.float32).copy(); em = np.ascontiguousarray(w[3], np.int32)
            ek = np.ascontiguousarray(w[4], np.int32); eo = np.ascontiguousarray(w[5], np.float32)
            ew = np.ascontiguousarray(w[6], np.float32); K4 = np.ascontiguousarray(w[7], np.float32)
            erase = np.zeros(max(len(em), 1), np.uint8)
            keep.append((kp, kfix, mp, em, ek, eo, ew, K4, erase))
            W.n_kf, W.kf_pose7, W.kf_fixed, W.n_mp, W.mp_pos3 = len(kfix), kp.ctypes.data, kfix.ctypes.data, len(mp), mp.ctypes.data
            W.n_edges, W.e_mp, W.e_kf, W.e_obs, W.e_inv_sigma2 = len(em), em.ctypes.data, ek.ctypes.data, eo.ctypes.data, ew.ctypes.data
            W.K4, W.stop_flag, W.erase_out = K4.ctypes.data, None, erase.ctypes.data
        import time
        t0 = time.perf_counter()
        rc = self._lib.rumi_local_ba_batch(self._h, len(windows), C.cast(arr, C.c_void_p), int(n_workers))
        self.last_call_s = time.perf_counter() - t0         # the C entry alone (the mirror above copies every array of every window)
        capi.check(rc)
        return [(np.array(W.stats, np.int32), k[0], k[2], k[8][:len(k[3])]) for W, k in zip(arr, keep)]

    def BundleAdjustment(self, kf_pose, kf_fixed, mp_pos, e_mp, e_kf, e_obs, e_inv_sigma2, K4, n_iterations=5, robust=True, stop_flag=None):
        """Optimizer::BundleAdjustment (global BA): returns (stats[4], kf_pose, mp_pos)."""
        kp = np.ascontiguousarray(kf_pose, np.float32).copy(); kfix = np.ascontiguousarray(kf_fixed, np.uint8)
        mp = np.ascontiguousarray(mp_pos, np.float32).copy(); em = np.ascontiguousarray(e_mp, np.int32)
        ek = np.ascontiguousarray(e_kf, np.int32); eo = np.ascontiguousarray(e_obs, np.float32)
        ew = np.ascontiguousarray(e_inv_sigma2, np.float32); K4 = np.ascontiguousarray(K4, np.float32)
        stats = np.zeros(4, np.int32)
        sp = capi.ptr(stop_flag) if stop_flag is not None else None
        capi.check(self._lib.rumi_bundle_adjustment(self._h, len(kfix), capi.ptr(kp), capi.ptr(kfix), len(mp), capi.ptr(mp), len(em), capi.ptr(em),
                                                    capi.ptr(ek), capi.ptr(eo), capi.ptr(ew), capi.ptr(K4), sp, int(n_iterations), int(robust), capi.ptr(stats)))
        return stats, kp, mp

    def MergeBundleAdjustment(self, *args, **kw):
        """Optimizer::LocalBundleAdjustment(pMainKF, vpAdjustKF, vpFixedKF, pbStopFlag): the two-pass merge-window BA."""
        return self.LocalBundleAdjustment(*args, merge=True, **kw)

    def ComputeInliersNum(self, pair_start, pair_denominator, S_c1w2, S_c2w1, K4_1, K4_2, X1, X2, kp1, kp2, sigma2_1, sigma2_2, edge1, edge2):
        """Sim3Solver::ComputeInliersNum on flat arrays (include/rumi_opt.h).  Returns (median ratio, ratio per pair, inlier flags)."""
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        ps = np.ascontiguousarray(pair_start, np.int32); pd = np.ascontiguousarray(pair_denominator, np.int32)
        A = np.ascontiguousarray(S_c1w2, np.float64); B = np.ascontiguousarray(S_c2w1, np.float64)
        arrs = [f32(K4_1), f32(K4_2), f32(X1), f32(X2), f32(kp1), f32(kp2), f32(sigma2_1), f32(sigma2_2), np.ascontiguousarray(edge1, np.uint8),
                np.ascontiguousarray(edge2, np.uint8)]
        n_pairs, total = len(ps) - 1, int(ps[-1])
        inl = np.zeros(max(total, 1), np.uint8); ratio = np.zeros(max(n_pairs, 1), np.float32); med = C.c_float()
        capi.check(self._lib.rumi_sim3_inliers(self._h, n_pairs, capi.ptr(ps), capi.ptr(pd), capi.ptr(A), capi.ptr(B), *[capi.ptr(a) for a in arrs],
                                               capi.ptr(inl), capi.ptr(ratio), C.byref(med)))
        return med.value, ratio[:n_pairs], inl[:total]

    def Sim3Ransac(self, X3Dc1, X3Dc2, sigma2_1, sigma2_2, K4_1, K4_2, triples, fix_scale=False, score=None, want_inliers=True):
        """The hypotheses of one block of Sim3Solver::iterate (R/lib_src/Sim3Solver.cc:159-404; include/rumi_opt.h, rumi_sim3_ransac).
        triples [n_hyp,3] = the minimal sets.  score: None or a dict with the ComputeInliersNum data (pair_start, pair_denominator, S_c1w1, S_c2w2,
        S_kf1w, S_kf2w, K4_1, K4_2, X1, X2, kp1, kp2, sigma2_1, sigma2_2, edge1, edge2).
        Returns dict(R [h,3,3], t [h,3], s [h], valid [h], n_inliers [h], inliers [h,n] or None, ratio [h,n_pairs] or None, median [h] or None)."""
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        X1, X2, s1, s2, K1, K2 = f32(X3Dc1).reshape(-1, 3), f32(X3Dc2).reshape(-1, 3), f32(sigma2_1), f32(sigma2_2), f32(K4_1), f32(K4_2)
        tri = np.ascontiguousarray(triples, np.int32).reshape(-1, 3)
        n, H = len(X1), len(tri)
        T = np.zeros((max(H, 1), 16), np.float32); nin = np.zeros(max(H, 1), np.int32)
        inl = np.zeros((max(H, 1), n), np.uint8) if want_inliers else None
        sc, keep, ratio, med = None, [], None, None
        if score is not None:
            f64 = lambda a: np.ascontiguousarray(a, np.float64)
            u8 = lambda a: np.ascontiguousarray(a, np.uint8)
            ps, pd = np.ascontiguousarray(score["pair_start"], np.int32), np.ascontiguousarray(score["pair_denominator"], np.int32)
            keep = [ps, pd, f64(score["S_c1w1"]), f64(score["S_c2w2"]), f64(score["S_kf1w"]), f64(score["S_kf2w"]), f32(score["K4_1"]), f32(score["K4_2"]),
                    f32(score["X1"]), f32(score["X2"]), f32(score["kp1"]), f32(score["kp2"]), f32(score["sigma2_1"]), f32(score["sigma2_2"]),
                    u8(score["edge1"]), u8(score["edge2"])]
            sc = RumiSim3ScoreSet(len(ps) - 1, *[a.ctypes.data_as(C.c_void_p) for a in keep])
            ratio = np.zeros((max(H, 1), len(ps) - 1), np.float32); med = np.zeros(max(H, 1), np.float32)
        P = capi.ptr
        capi.check(self._lib.rumi_sim3_ransac(self._h, n, P(X1), P(X2), P(s1), P(s2), P(K1), P(K2), int(bool(fix_scale)), H, P(tri),
                                              C.byref(sc) if sc is not None else None, P(T), P(nin), P(inl) if inl is not None else None,
                                              P(ratio) if ratio is not None else None, P(med) if med is not None else None))
        T = T[:H]
        return dict(R=T[:, :9].reshape(-1, 3, 3).copy(), t=T[:, 9:12].copy(), s=T[:, 12].copy(), valid=T[:, 13] != 0, n_inliers=nin[:H].copy(),
                    inliers=inl[:H].astype(bool) if inl is not None else None, ratio=ratio[:H] if ratio is not None else None,
                    median=med[:H] if med is not None else None)

    def OptimizeSim3(self, S8, P1c, P2c, obs1, obs2, inv_sigma2_1, inv_sigma2_2, K4_1, K4_2, th2=10.0, fix_scale=False, robust_first_pass=True,
                     pair_of=None, S_c1w=None, S_c2w=None, skip12=None, skip21=None):
        """Optimizer::OptimizeSim3 (robust first pass, no key-frame transforms) and, with pair_of / S_c1w / S_c2w and
        robust_first_pass=False, Optimizer::OptimizeCloudSim3, on flat arrays (include/rumi_opt.h).
        Returns (nIn, nBad, early, S8, status[n])."""
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        u8 = lambda a: None if a is None else np.ascontiguousarray(a, np.uint8)
        P1c, P2c, obs1, obs2, w1, w2, K1, K2 = (f32(a) for a in (P1c, P2c, obs1, obs2, inv_sigma2_1, inv_sigma2_2, K4_1, K4_2))
        n = len(w1)
        S = np.ascontiguousarray(S8, np.float64).copy()
        world = S_c1w is not None
        po = np.ascontiguousarray(pair_of, np.int32) if world else None
        A = np.ascontiguousarray(S_c1w, np.float64).reshape(-1, 8) if world else None
        B = np.ascontiguousarray(S_c2w, np.float64).reshape(-1, 8) if world else None
        s12, s21 = u8(skip12), u8(skip21)
        status = np.zeros(max(n, 1), np.uint8); res = np.zeros(3, np.int32)
        P = lambda a: None if a is None else capi.ptr(a)
        capi.check(self._lib.rumi_optimize_sim3(self._h, n, P(po), len(A) if world else 0, P(A), P(B), P(P1c), P(P2c), P(obs1), P(obs2), P(w1), P(w2),
                                                P(s12), P(s21), P(K1), P(K2), float(th2), int(fix_scale), int(robust_first_pass), P(S), P(status), P(res)))
        return int(res[0]), int(res[1]), bool(res[2]), S, status[:n]

    def set_profiling(self, on=True):
        capi.check(self._lib.rumi_opt_set_profiling(self._h, int(on)))

    def kernel_ms(self):
        """Per-kernel device time of the last bundle adjustment run with profiling on: dict(hpp, syrk, solve in ms over the call, trials)."""
        ms = np.zeros(4, np.float32)
        capi.check(self._lib.rumi_opt_kernel_ms(self._h, capi.ptr(ms)))
        return dict(hpp=float(ms[0]), syrk=float(ms[1]), solve=float(ms[2]), trials=int(ms[3]))

    def stage_ms(self):
        ms = np.zeros(8, np.float32)
        capi.check(self._lib.rumi_opt_stage_ms(self._h, capi.ptr(ms)))
        return ms
