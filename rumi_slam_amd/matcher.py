"""Host-side mirror of ``ORB_SLAM3::ORBmatcher`` (R/include/cloud_edge_slam_lib/ORBmatcher.h:36-103) over the C ABI in
include/rumi_match.h.  Frames / key-frames are passed as ``FrameView`` (the flat arrays the matchers read), map
points as integer ids into caller-side arrays (-1 = NULL)."""
import ctypes as C

import numpy as np

from . import capi
from .capi import KP_DTYPE

TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30


class RumiFrameFeatures(C.Structure):
    _fields_ = [("n", C.c_int32), ("keys_un", C.c_void_p), ("desc", C.c_void_p), ("min_x", C.c_float), ("min_y", C.c_float),
                ("max_x", C.c_float), ("max_y", C.c_float), ("scale_factors", C.c_void_p), ("nlevels", C.c_int32)]


class RumiFeatureVector(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("node_ids", C.c_void_p), ("offsets", C.c_void_p), ("indices", C.c_void_p)]


MATCH_SYMBOLS = ["rumi_descriptor_distance", "rumi_match_create", "rumi_match_destroy", "rumi_search_by_projection_mappoints",
                 "rumi_search_by_projection_frame", "rumi_search_by_bow", "rumi_search_by_bow_kf", "rumi_search_by_projection_sim3",
                 "rumi_search_by_projection_reloc", "rumi_search_for_initialization", "rumi_search_for_triangulation", "rumi_fuse_candidates", "rumi_search_by_sim3", "rumi_frame_is_in_frustum", "rumi_search_by_bow_batch", "rumi_match_bruteforce_batch_device",
                 "rumi_match_bruteforce_batch_device_strided", "rumi_match_bruteforce_ring_device"]


def _lib():
    L = capi.lib()
    if getattr(L, "_match_ready", False):
        return L
    vp, i32, f32 = C.c_void_p, C.c_int32, C.c_float
    L.rumi_descriptor_distance.argtypes = [vp, vp]
    L.rumi_match_create.argtypes = [i32, i32, i32, C.POINTER(vp)]
    L.rumi_match_destroy.argtypes = [vp]
    L.rumi_match_destroy.restype = None
    L.rumi_search_by_projection_mappoints.argtypes = [vp, C.POINTER(RumiFrameFeatures), i32] + [vp] * 9 + [f32, i32, f32, f32, vp, C.POINTER(i32)]
    L.rumi_search_by_projection_frame.argtypes = [vp, C.POINTER(RumiFrameFeatures), vp, vp, vp, i32, vp, vp, i32, vp, vp, vp, f32, i32, vp, C.POINTER(i32)]
    L.rumi_search_by_bow.argtypes = [vp, C.POINTER(RumiFrameFeatures), C.POINTER(RumiFeatureVector), vp, i32, vp,
                                     C.POINTER(RumiFrameFeatures), C.POINTER(RumiFeatureVector), f32, i32, vp, C.POINTER(i32)]
    L.rumi_search_by_bow_kf.argtypes = [vp, C.POINTER(RumiFrameFeatures), C.POINTER(RumiFeatureVector), vp, C.POINTER(RumiFrameFeatures),
                                        C.POINTER(RumiFeatureVector), vp, i32, vp, f32, i32, vp, C.POINTER(i32)]
    L.rumi_search_by_projection_sim3.argtypes = [vp, C.POINTER(RumiFrameFeatures), f32, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, f32, i32,
                                                 vp, C.POINTER(i32)]
    L.rumi_search_by_projection_reloc.argtypes = [vp, C.POINTER(RumiFrameFeatures), f32, vp, vp, vp, vp, i32, vp, i32, vp, vp, vp, vp, vp, f32,
                                                  i32, i32, vp, C.POINTER(i32)]
    L.rumi_search_for_initialization.argtypes = [vp, vp, vp, vp, i32, f32, i32, vp, vp]
    L.rumi_search_for_triangulation.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp]
    L.rumi_fuse_candidates.argtypes = [vp, vp, f32, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, f32, i32, vp]
    L.rumi_search_by_sim3.argtypes = [vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, f32, vp, vp]
    L.rumi_search_local_points.argtypes = [vp, C.POINTER(RumiFrameFeatures), vp, vp, vp, vp, f32, i32, f32, i32] + [vp] * 7 + [f32, i32, f32, f32] + [vp] * 6 + [C.POINTER(i32), vp, C.POINTER(i32)]
    L.rumi_frame_is_in_frustum.argtypes = [vp, vp, vp, vp, vp, f32, f32, f32, f32, f32, i32, f32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.rumi_match_bruteforce_batch_device.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]
    L.rumi_search_by_bow_batch.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, C.c_float, i32, vp, vp]
    L.rumi_match_bruteforce_batch_device_strided.argtypes = [vp, vp, vp, vp, i32, C.c_int64, C.c_int64, i32, i32, vp, vp, vp, vp]
    L.rumi_match_bruteforce_ring_device.argtypes = [vp, vp, i32, C.c_int64, i32, i32, vp, vp, vp, vp]
    L._match_ready = True
    return L


class FrameView:
    """The part of a Frame / KeyFrame the matchers read (keeps the arrays alive)."""

    def __init__(self, keys_un, desc, width, height, scale_factors, min_x=0.0, min_y=0.0):
        self.keys = np.ascontiguousarray(keys_un, KP_DTYPE)
        self.desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        self.sf = np.ascontiguousarray(scale_factors, np.float32)
        assert len(self.keys) == len(self.desc)
        self.bounds = (float(min_x), float(min_y), float(width), float(height))
        self.c = RumiFrameFeatures(len(self.keys), capi.ptr(self.keys), capi.ptr(self.desc), self.bounds[0], self.bounds[1],
                                   self.bounds[2], self.bounds[3], capi.ptr(self.sf), len(self.sf))

    @property
    def n(self):
        return len(self.keys)


class FeatureVector:
    """DBoW2::FeatureVector as CSR (node ids ascending = std::map order)."""

    def __init__(self, mapping):
        nodes = sorted(mapping)
        self.node_ids = np.array(nodes, np.uint32)
        self.offsets = np.zeros(len(nodes) + 1, np.int32)
        idx = []
        for k, nid in enumerate(nodes):
            idx.extend(mapping[nid])
            self.offsets[k + 1] = len(idx)
        self.indices = np.array(idx, np.uint32)
        self.c = RumiFeatureVector(len(nodes), capi.ptr(self.node_ids), capi.ptr(self.offsets), capi.ptr(self.indices))

    @classmethod
    def from_csr(cls, node_ids, offsets, indices):
        """From the CSR arrays ``ORBVocabulary.transform`` / ``assemble`` return."""
        fv = cls.__new__(cls)
        fv.node_ids = np.ascontiguousarray(node_ids, np.uint32); fv.offsets = np.ascontiguousarray(offsets, np.int32)
        fv.indices = np.ascontiguousarray(indices, np.uint32)
        fv.c = RumiFeatureVector(len(fv.node_ids), capi.ptr(fv.node_ids), capi.ptr(fv.offsets), capi.ptr(fv.indices))
        return fv


def DescriptorDistance(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return _lib().rumi_descriptor_distance(capi.ptr(a), capi.ptr(b))


class ORBmatcher:
    def __init__(self, nnratio=0.6, checkOri=True, max_features=8192, max_queries=16384, device=-1):
        self._lib = _lib()
        self.mfNNratio, self.mbCheckOrientation = float(nnratio), bool(checkOri)
        self._h = C.c_void_p()
        capi.check(self._lib.rumi_match_create(max_features, max_queries, device, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.rumi_match_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def SearchByProjection_MapPoints(self, F, mp, frame_mp, th=1.0, bFarPoints=False, thFarPoints=50.0):
        """mp: dict of arrays track_in_view,u8 proj_x,proj_y,view_cos,track_depth f32 scale_level,obs i32 is_bad u8 desc[n,32]."""
        n = len(mp["proj_x"])
        a = {k: np.ascontiguousarray(mp[k], t) for k, t in [("track_in_view", np.uint8), ("proj_x", np.float32), ("proj_y", np.float32),
                                                             ("scale_level", np.int32), ("view_cos", np.float32), ("track_depth", np.float32),
                                                             ("is_bad", np.uint8), ("desc", np.uint8), ("obs", np.int32)]}
        frame_mp = np.ascontiguousarray(frame_mp, np.int32).copy()
        nm = C.c_int32()
        capi.check(self._lib.rumi_search_by_projection_mappoints(
            self._h, C.byref(F.c), n, capi.ptr(a["track_in_view"]), capi.ptr(a["proj_x"]), capi.ptr(a["proj_y"]), capi.ptr(a["scale_level"]),
            capi.ptr(a["view_cos"]), capi.ptr(a["track_depth"]), capi.ptr(a["is_bad"]), capi.ptr(a["desc"]), capi.ptr(a["obs"]),
            float(th), int(bFarPoints), float(thFarPoints), self.mfNNratio, capi.ptr(frame_mp), C.byref(nm)))
        return nm.value, frame_mp

    def SearchLocalPoints(self, F, Rcw9, tcw3, Ow3, K4, log_sf, nlevels, pts, frame_mp, th=1.0, bFarPoints=False, thFarPoints=50.0, cos_limit=0.5):
        """Tracking::SearchLocalPoints' frustum test + SearchByProjection(F, local points) in one call (rumi_search_local_points).
        pts: dict pos, normal [n,3], min_dist, max_dist [n], desc [n,32], obs [n], skip [n] (mnLastFrameSeen == frame id or isBad()).
        Returns (nToMatch, nmatches, frame_mp, fields) with fields = the six per-point arrays isInFrustum writes."""
        n = len(pts["max_dist"])
        R, t, O, K = _f32(Rcw9), _f32(tcw3), _f32(Ow3), _f32(K4)
        a = dict(pos=_f32(pts["pos"]), normal=_f32(pts["normal"]), mn=_f32(pts["min_dist"]), mx=_f32(pts["max_dist"]),
                 desc=np.ascontiguousarray(pts["desc"], np.uint8), obs=np.ascontiguousarray(pts["obs"], np.int32),
                 skip=np.ascontiguousarray(pts["skip"], np.uint8))
        out = dict(track_in_view=np.zeros(n, np.uint8), proj_x=np.zeros(n, np.float32), proj_y=np.zeros(n, np.float32),
                   scale_level=np.zeros(n, np.int32), view_cos=np.zeros(n, np.float32), track_depth=np.zeros(n, np.float32))
        frame_mp = np.ascontiguousarray(frame_mp, np.int32).copy()
        nto, nm = C.c_int32(), C.c_int32()
        capi.check(self._lib.rumi_search_local_points(
            self._h, C.byref(F.c), capi.ptr(R), capi.ptr(t), capi.ptr(O), capi.ptr(K), float(log_sf), int(nlevels), float(cos_limit), n, capi.ptr(a["skip"]),
            capi.ptr(a["pos"]), capi.ptr(a["normal"]), capi.ptr(a["mn"]), capi.ptr(a["mx"]), capi.ptr(a["desc"]), capi.ptr(a["obs"]), float(th),
            int(bFarPoints), float(thFarPoints), self.mfNNratio, capi.ptr(out["track_in_view"]), capi.ptr(out["proj_x"]), capi.ptr(out["proj_y"]),
            capi.ptr(out["scale_level"]), capi.ptr(out["view_cos"]), capi.ptr(out["track_depth"]), C.byref(nto), capi.ptr(frame_mp), C.byref(nm)))
        return nto.value, nm.value, frame_mp, out

    def SearchByProjection_Frame(self, Cur, Tcw7, K4, last_keys, last_mp, last_outlier, mp_pos, mp_desc, mp_obs, cur_mp, th):
        Tcw7 = np.ascontiguousarray(Tcw7, np.float32); K4 = np.ascontiguousarray(K4, np.float32)
        last_keys = np.ascontiguousarray(last_keys, KP_DTYPE); last_mp = np.ascontiguousarray(last_mp, np.int32)
        last_outlier = np.ascontiguousarray(last_outlier, np.uint8); mp_pos = np.ascontiguousarray(mp_pos, np.float32)
        mp_desc = np.ascontiguousarray(mp_desc, np.uint8); mp_obs = np.ascontiguousarray(mp_obs, np.int32)
        cur_mp = np.ascontiguousarray(cur_mp, np.int32).copy()
        nm = C.c_int32()
        capi.check(self._lib.rumi_search_by_projection_frame(
            self._h, C.byref(Cur.c), capi.ptr(Tcw7), capi.ptr(K4), capi.ptr(last_keys), len(last_keys), capi.ptr(last_mp),
            capi.ptr(last_outlier), len(mp_obs), capi.ptr(mp_pos), capi.ptr(mp_desc), capi.ptr(mp_obs), float(th),
            int(self.mbCheckOrientation), capi.ptr(cur_mp), C.byref(nm)))
        return nm.value, cur_mp

    def SearchForInitialization(self, F1, F2, prev_matched, window_size=100):
        """Returns (nmatches, vnMatches12, updated vbPrevMatched) — ORBmatcher.cc:581-680."""
        pm = np.ascontiguousarray(prev_matched, np.float32).copy()
        m12 = np.full(F1.n, -1, np.int32)
        nm = C.c_int32()
        capi.check(self._lib.rumi_search_for_initialization(self._h, C.byref(F1.c), C.byref(F2.c), capi.ptr(pm), int(window_size), self.mfNNratio,
                                                            int(self.mbCheckOrientation), capi.ptr(m12), C.byref(nm)))
        return nm.value, m12, pm

    def SearchByBoW(self, KF, kf_fv, kf_mp, mp_bad, F, f_fv):
        kf_mp = np.ascontiguousarray(kf_mp, np.int32); mp_bad = np.ascontiguousarray(mp_bad, np.uint8)
        matches = np.full(F.n, -1, np.int32)
        nm = C.c_int32()
        capi.check(self._lib.rumi_search_by_bow(self._h, C.byref(KF.c), C.byref(kf_fv.c), capi.ptr(kf_mp), len(mp_bad), capi.ptr(mp_bad),
                                                C.byref(F.c), C.byref(f_fv.c), self.mfNNratio, int(self.mbCheckOrientation),
                                                capi.ptr(matches), C.byref(nm)))
        return nm.value, matches


def SearchByBoW_batch(m, KFs, kf_fvs, kf_mps, mp_bads, F, f_fv):
    """SearchByBoW(KF_k, F) for K candidate key-frames against one frame in one launch (the walk over the relocalisation candidates,
    Tracking.cc:3240-3260).  KFs / kf_fvs: lists of FrameFeatures / FeatureVector; kf_mps[k], mp_bads[k]: per key-frame arrays as for
    ``ORBmatcher.SearchByBoW``.  Returns (nmatches [K], matches [K, F.n])."""
    K = len(KFs)
    kfa = (RumiFrameFeatures * K)(*[k.c for k in KFs])
    fva = (RumiFeatureVector * K)(*[v.c for v in kf_fvs])
    mps = [np.ascontiguousarray(a, np.int32) for a in kf_mps]
    bads = [np.ascontiguousarray(a, np.uint8) for a in mp_bads]
    mpp = (C.c_void_p * K)(*[a.ctypes.data for a in mps])
    bdp = (C.c_void_p * K)(*[a.ctypes.data for a in bads])
    nmp = np.array([len(a) for a in bads], np.int32)
    matches = np.full((K, F.n), -1, np.int32)
    nm = np.zeros(K, np.int32)
    capi.check(m._lib.rumi_search_by_bow_batch(m._h, K, kfa, fva, mpp, capi.ptr(nmp), bdp, C.byref(F.c), C.byref(f_fv.c), m.mfNNratio,
                                               int(m.mbCheckOrientation), capi.ptr(matches), capi.ptr(nm)))
    return nm, matches


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


def _u8(a):
    return np.ascontiguousarray(a, np.uint8)


def SearchByBoW_KF(m, KF1, fv1, kf1_mp, KF2, fv2, kf2_mp, mp_bad):
    """SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12): returns (nmatches, matches12[KF1.n] = KF2 feature index or -1)."""
    k1 = np.ascontiguousarray(kf1_mp, np.int32); k2 = np.ascontiguousarray(kf2_mp, np.int32); mb = _u8(mp_bad)
    out = np.full(KF1.n, -1, np.int32)
    nm = C.c_int32()
    capi.check(m._lib.rumi_search_by_bow_kf(m._h, C.byref(KF1.c), C.byref(fv1.c), capi.ptr(k1), C.byref(KF2.c), C.byref(fv2.c), capi.ptr(k2),
                                            len(mb), capi.ptr(mb), m.mfNNratio, int(m.mbCheckOrientation), capi.ptr(out), C.byref(nm)))
    return nm.value, out


def SearchForTriangulation(m, KF1, fv1, kf1_mp, KF2, fv2, kf2_mp, F12, epipole2, only_stereo=False, coarse=False):
    """ORBmatcher::SearchForTriangulation (mono): returns (nmatches, vMatchedPairs as an [k,2] array of (idx1, idx2))."""
    k1 = np.ascontiguousarray(kf1_mp, np.int32); k2 = np.ascontiguousarray(kf2_mp, np.int32)
    F = _f32(F12).ravel(); ep = _f32(epipole2)
    out = np.full(KF1.n, -1, np.int32)
    nm = C.c_int32()
    capi.check(m._lib.rumi_search_for_triangulation(m._h, C.byref(KF1.c), C.byref(fv1.c), capi.ptr(k1), C.byref(KF2.c), C.byref(fv2.c), capi.ptr(k2),
                                                    capi.ptr(F), capi.ptr(ep), int(only_stereo), int(coarse), int(m.mbCheckOrientation),
                                                    capi.ptr(out), C.byref(nm)))
    idx = np.nonzero(out >= 0)[0]
    return nm.value, np.stack([idx, out[idx]], 1).astype(np.int64)


def SearchBySim3(m, KF1, KF2, K4, log_sf, side1, side2, th):
    """ORBmatcher::SearchBySim3.  side*: dict skip u8, pc [n,3] (point in the OTHER camera), min_dist, max_dist, desc [n,32].
    Returns (nFound, match12[KF1.n])."""
    def prep(d):
        return [_u8(d["skip"]), _f32(d["pc"]), _f32(d["min_dist"]), _f32(d["max_dist"]), _u8(d["desc"])]
    a, b = prep(side1), prep(side2)
    K = _f32(K4)
    out = np.full(KF1.n, -1, np.int32)
    nf = C.c_int32()
    capi.check(m._lib.rumi_search_by_sim3(m._h, C.byref(KF1.c), C.byref(KF2.c), capi.ptr(K), float(log_sf), *[capi.ptr(x) for x in a],
                                          *[capi.ptr(x) for x in b], float(th), capi.ptr(out), C.byref(nf)))
    return nf.value, out


def FuseCandidates(m, KF, log_sf, Tcw7, Ow3, K4, pts, th, check_reprojection=True):
    """Search half of ORBmatcher::Fuse: best key-frame feature per map point (-1 none).  pts as for SearchByProjection_Sim3."""
    a = dict(skip=_u8(pts["skip"]), pos=_f32(pts["pos"]), normal=_f32(pts["normal"]), mn=_f32(pts["min_dist"]), mx=_f32(pts["max_dist"]),
             desc=_u8(pts["desc"]))
    n = len(a["skip"])
    out = np.full(n, -1, np.int32)
    T, O, K = _f32(Tcw7), _f32(Ow3), _f32(K4)
    capi.check(m._lib.rumi_fuse_candidates(m._h, C.byref(KF.c), float(log_sf), capi.ptr(T), capi.ptr(O), capi.ptr(K), n, capi.ptr(a["skip"]),
                                           capi.ptr(a["pos"]), capi.ptr(a["normal"]), capi.ptr(a["mn"]), capi.ptr(a["mx"]), capi.ptr(a["desc"]),
                                           float(th), int(check_reprojection), capi.ptr(out)))
    return out


def SearchByProjection_Sim3(m, KF, log_sf, Tcw7, Ow3, K4, pts, matched, th, ratio_hamming, explicit_invz=False):
    """pts: dict skip u8, pos [n,3], normal [n,3], min_dist, max_dist f32, desc [n,32].  matched in/out (-1 free)."""
    a = dict(skip=_u8(pts["skip"]), pos=_f32(pts["pos"]), normal=_f32(pts["normal"]), mn=_f32(pts["min_dist"]), mx=_f32(pts["max_dist"]),
             desc=_u8(pts["desc"]))
    matched = np.ascontiguousarray(matched, np.int32).copy()
    T, O, K = _f32(Tcw7), _f32(Ow3), _f32(K4)
    nm = C.c_int32()
    capi.check(m._lib.rumi_search_by_projection_sim3(m._h, C.byref(KF.c), float(log_sf), capi.ptr(T), capi.ptr(O), capi.ptr(K), len(a["skip"]),
                                                     capi.ptr(a["skip"]), capi.ptr(a["pos"]), capi.ptr(a["normal"]), capi.ptr(a["mn"]),
                                                     capi.ptr(a["mx"]), capi.ptr(a["desc"]), int(th), float(ratio_hamming), int(explicit_invz),
                                                     capi.ptr(matched), C.byref(nm)))
    return nm.value, matched


def SearchByProjection_Reloc(m, Cur, log_sf, Tcw7, Ow3, K4, kf_keys, kf_mp, pts, cur_mp, th, orb_dist):
    a = dict(skip=_u8(pts["skip"]), pos=_f32(pts["pos"]), mn=_f32(pts["min_dist"]), mx=_f32(pts["max_dist"]), desc=_u8(pts["desc"]))
    kk = np.ascontiguousarray(kf_keys, KP_DTYPE); km = np.ascontiguousarray(kf_mp, np.int32)
    cur_mp = np.ascontiguousarray(cur_mp, np.int32).copy()
    T, O, K = _f32(Tcw7), _f32(Ow3), _f32(K4)
    nm = C.c_int32()
    capi.check(m._lib.rumi_search_by_projection_reloc(m._h, C.byref(Cur.c), float(log_sf), capi.ptr(T), capi.ptr(O), capi.ptr(K), capi.ptr(kk),
                                                      len(kk), capi.ptr(km), len(a["skip"]), capi.ptr(a["skip"]), capi.ptr(a["pos"]),
                                                      capi.ptr(a["mn"]), capi.ptr(a["mx"]), capi.ptr(a["desc"]), float(th), int(orb_dist),
                                                      int(m.mbCheckOrientation), capi.ptr(cur_mp), C.byref(nm)))
    return nm.value, cur_mp


def isInFrustum(m, Rcw9, tcw3, Ow3, K4, w, h, log_sf, nlevels, cos_limit, pts):
    """Frame::isInFrustum for all points; returns the dict SearchByProjection_MapPoints takes (without desc / obs / is_bad)."""
    n = len(pts["max_dist"])
    R, t, O, K = _f32(Rcw9), _f32(tcw3), _f32(Ow3), _f32(K4)
    a = dict(pos=_f32(pts["pos"]), normal=_f32(pts["normal"]), mn=_f32(pts["min_dist"]), mx=_f32(pts["max_dist"]))
    out = dict(track_in_view=np.zeros(n, np.uint8), proj_x=np.zeros(n, np.float32), proj_y=np.zeros(n, np.float32),
               scale_level=np.zeros(n, np.int32), view_cos=np.zeros(n, np.float32), track_depth=np.zeros(n, np.float32))
    capi.check(m._lib.rumi_frame_is_in_frustum(m._h, capi.ptr(R), capi.ptr(t), capi.ptr(O), capi.ptr(K), 0.0, 0.0, float(w), float(h), float(log_sf),
                                               int(nlevels), float(cos_limit), n, capi.ptr(a["pos"]), capi.ptr(a["normal"]), capi.ptr(a["mn"]),
                                               capi.ptr(a["mx"]), capi.ptr(out["track_in_view"]), capi.ptr(out["proj_x"]), capi.ptr(out["proj_y"]),
                                               capi.ptr(out["scale_level"]), capi.ptr(out["view_cos"]), capi.ptr(out["track_depth"])))
    return out


def bruteforce_batch(desc_q, counts_q, desc_t, counts_t, stream=None):
    """desc_*: torch u8 CUDA [B,cap,32] (frames may be strided views, e.g. of the per-frame records); counts_*: torch i32 CUDA [B,2]
    (extractor counts, any frame stride).  Returns best_idx, best, second [B,cap]."""
    import torch
    B, cap, _ = desc_q.shape
    assert desc_q.stride(1) == 32 and desc_q.stride(2) == 1 and desc_t.stride(1) == 32 and desc_t.stride(2) == 1
    assert counts_q.stride(0) == counts_t.stride(0) or B == 1
    out = [torch.empty((B, cap), dtype=torch.int32, device=desc_q.device) for _ in range(3)]
    st = stream if stream is not None else torch.cuda.current_stream(desc_q.device)
    capi.check(_lib().rumi_match_bruteforce_batch_device_strided(desc_q.data_ptr(), counts_q.data_ptr(), desc_t.data_ptr(), counts_t.data_ptr(),
                                                                 counts_q.stride(0), desc_q.stride(0) if B > 1 else 32 * cap,
                                                                 desc_t.stride(0) if B > 1 else 32 * cap, cap, B, out[0].data_ptr(),
                                                                 out[1].data_ptr(), out[2].data_ptr(), st.cuda_stream))
    return out


def bruteforce_ring(desc, counts, stream=None, out=None):
    """Frame i against frame i + 1 of one buffer, the last against the first, in one launch (the consecutive-frame matching of a rumination
    queue).  desc: torch u8 CUDA [B,cap,32] (frames may be a strided view); counts: torch i32 CUDA [B,2].  Returns best_idx, best, second
    [B,cap] (out: three preallocated int32 tensors of that shape)."""
    import torch
    B, cap, _ = desc.shape
    assert desc.stride(1) == 32 and desc.stride(2) == 1
    if out is None:
        out = [torch.empty((B, cap), dtype=torch.int32, device=desc.device) for _ in range(3)]
    st = stream if stream is not None else torch.cuda.current_stream(desc.device)
    capi.check(_lib().rumi_match_bruteforce_ring_device(desc.data_ptr(), counts.data_ptr(), counts.stride(0), desc.stride(0) if B > 1 else 32 * cap, cap, B,
                                                        out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), st.cuda_stream))
    return out
