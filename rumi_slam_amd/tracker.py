"""Host-side mirror of include/rumi_track.h: one device-resident Tracking step (TrackWithMotionModel + TrackLocalMap data path,
R/lib_src/Tracking.cc:2441-2607, 2996-3055) behind one call."""
import ctypes as C

import time

import numpy as np

from . import capi
from .capi import KP_DTYPE, RumiOrbConfig


class RumiTrackPoints(C.Structure):
    _fields_ = [("n", C.c_int32), ("pos", C.c_void_p), ("normal", C.c_void_p), ("min_dist", C.c_void_p), ("max_dist", C.c_void_p),
                ("desc", C.c_void_p), ("obs", C.c_void_p), ("bad", C.c_void_p), ("local", C.c_void_p),
                ("stale_in_view", C.c_void_p), ("stale_proj", C.c_void_p)]       # optional: points["stale_in_view"] [n] u8, points["stale_proj"] [n,5] f32


def _stale(points, a):
    """The optional stale-mbTrackInView arrays of include/rumi_track.h (RumiTrackPoints.stale_in_view / stale_proj) as two pointers (None, None when absent)."""
    if points.get("stale_in_view") is None:
        return None, None
    a["stale_in"] = np.ascontiguousarray(points["stale_in_view"], np.uint8)
    a["stale_proj"] = np.ascontiguousarray(points["stale_proj"], np.float32).reshape(-1, 5)
    assert len(a["stale_in"]) == len(a["stale_proj"]) == len(points["obs"])
    return capi.ptr(a["stale_in"]), capi.ptr(a["stale_proj"])


class RumiTrackResult(C.Structure):
    _fields_ = [("n", C.c_int32), ("mono_index", C.c_int32), ("th_motion", C.c_int32), ("nmatches_motion", C.c_int32), ("ngood_motion", C.c_int32),
                ("nmatches_map", C.c_int32), ("n_to_match", C.c_int32), ("nmatches_local", C.c_int32), ("ngood_local", C.c_int32),
                ("matches_inliers", C.c_int32), ("Tcw_motion", C.c_float * 7), ("Tcw", C.c_float * 7), ("Rcw", C.c_float * 9), ("tcw", C.c_float * 3),
                ("Ow", C.c_float * 3)]


def _bind(L):
    if getattr(L, "_track_ready", False):
        return L
    vp, i32, f32 = C.c_void_p, C.c_int32, C.c_float
    L.rumi_track_create.argtypes = [C.POINTER(RumiOrbConfig), i32, i32, C.POINTER(vp)]
    L.rumi_track_destroy.argtypes = [vp]
    L.rumi_track_destroy.restype = None
    L.rumi_track_frame.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, i32, vp, vp, C.POINTER(RumiTrackPoints), f32, f32, i32, f32,
                                   vp, vp, i32, vp, vp, vp, vp, C.POINTER(RumiTrackResult)]
    L.rumi_track_extract.argtypes = [vp, vp, i32, i32, i32, vp, vp, i32, C.POINTER(i32), C.POINTER(i32)]
    L.rumi_track_motion.argtypes = [vp, vp, vp, vp, i32, vp, vp, C.POINTER(RumiTrackPoints), f32, vp, vp, C.POINTER(RumiTrackResult)]
    L.rumi_track_reference_keyframe.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, C.POINTER(RumiTrackPoints), f32, i32, vp, vp, vp, vp, vp,
                                                C.POINTER(RumiTrackResult)]
    L.rumi_track_local.argtypes = [vp, vp, vp, vp, C.POINTER(RumiTrackPoints), vp, f32, i32, f32, vp, vp, vp, C.POINTER(RumiTrackResult)]
    L.rumi_track_last_projections.argtypes = [vp, i32, vp]
    L.rumi_track_set_distortion.argtypes = [vp, vp, vp]
    L.rumi_track_undistorted.argtypes = [vp, vp, i32, vp]
    L.rumi_track_image_buffer.argtypes = [vp, i32, i32, C.POINTER(vp), C.POINTER(i32)]
    L._track_ready = True
    return L


class Tracker:
    """rumi_track_create / rumi_track_frame.  `points` of track(): dict pos, normal [n,3], min_dist, max_dist [n], desc [n,32], obs [n], bad [n], local [n]."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7, max_width=640, max_height=480, max_points=8192, device=-1,
                 blur_variant=0):
        self._lib = _bind(capi.lib())
        self.cfg = RumiOrbConfig(nfeatures, scale_factor, nlevels, ini_th, min_th, max_width, max_height, 1, device, 0, blur_variant)
        self.cap = nfeatures + 4 * nlevels + 64
        self._h = C.c_void_p()
        capi.check(self._lib.rumi_track_create(C.byref(self.cfg), int(max_points), int(device), C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.rumi_track_destroy(self._h)
            self._h = C.c_void_p()

    def track(self, img, K4, Tcw_pred7, last_keys, last_mp, last_outlier, points, th_motion=15.0, th_local=1.0, far_points=False, th_far_points=50.0):
        img = np.asarray(img, np.uint8)                      # rows may be strided (a view into a wider buffer); pixels of a row are contiguous
        assert img.ndim == 2 and img.strides[1] == 1
        h, w = img.shape
        K4 = np.ascontiguousarray(K4, np.float32); T = np.ascontiguousarray(Tcw_pred7, np.float32)
        lk = np.ascontiguousarray(last_keys, KP_DTYPE); lm = np.ascontiguousarray(last_mp, np.int32); lo = np.ascontiguousarray(last_outlier, np.uint8)
        n = len(points["obs"])
        a = dict(pos=np.ascontiguousarray(points["pos"], np.float32), normal=np.ascontiguousarray(points["normal"], np.float32),
                 mn=np.ascontiguousarray(points["min_dist"], np.float32), mx=np.ascontiguousarray(points["max_dist"], np.float32),
                 desc=np.ascontiguousarray(points["desc"], np.uint8), obs=np.ascontiguousarray(points["obs"], np.int32),
                 bad=np.ascontiguousarray(points["bad"], np.uint8), local=np.ascontiguousarray(points["local"], np.uint8))
        P = RumiTrackPoints(n, *(capi.ptr(a[k]) for k in ("pos", "normal", "mn", "mx", "desc", "obs", "bad", "local")), *_stale(points, a))
        keys = np.zeros(self.cap, KP_DTYPE); desc = np.zeros((self.cap, 32), np.uint8)
        mp_motion = np.full(self.cap, -1, np.int32); mp = np.full(self.cap, -1, np.int32); outl = np.zeros(self.cap, np.uint8)
        in_view = np.zeros(max(n, 1), np.uint8)
        res = RumiTrackResult()
        capi.check(self._lib.rumi_track_frame(self._h, capi.ptr(img), w, h, img.strides[0], capi.ptr(K4), capi.ptr(T), capi.ptr(lk), len(lk), capi.ptr(lm),
                                              capi.ptr(lo), C.byref(P), float(th_motion), float(th_local), int(far_points), float(th_far_points),
                                              capi.ptr(keys), capi.ptr(desc), self.cap, capi.ptr(mp_motion), capi.ptr(mp), capi.ptr(outl), capi.ptr(in_view),
                                              C.byref(res)))
        k = res.n
        out = {f: getattr(res, f) for f in ("n", "mono_index", "th_motion", "nmatches_motion", "ngood_motion", "nmatches_map", "n_to_match",
                                            "nmatches_local", "ngood_local", "matches_inliers")}
        for f in ("Tcw_motion", "Tcw", "Rcw", "tcw", "Ow"):
            out[f] = np.array(getattr(res, f), np.float32)
        out.update(keys=keys[:k].copy(), desc=desc[:k].copy(), frame_mp_motion=mp_motion[:k].copy(), frame_mp=mp[:k].copy(), outlier=outl[:k].copy(),
                   in_view=in_view[:n].copy())
        return out

    def image_buffer(self, w, h):
        """rumi_track_image_buffer: an [h, w] uint8 view of the tracker's pinned staging memory; a frame written there and passed to track() /
        extract() is uploaded without the staging copy."""
        buf, stride = C.c_void_p(), C.c_int32()
        capi.check(self._lib.rumi_track_image_buffer(self._h, int(w), int(h), C.byref(buf), C.byref(stride)))
        flat = np.ctypeslib.as_array((C.c_uint8 * (stride.value * h)).from_address(buf.value))
        return flat.reshape(h, stride.value)[:, :w]

    # ---- step-wise entries: one member function of Tracking per call, the frame resident in between (include/rumi_track.h) ----
    @staticmethod
    def _points(points, need_frustum):
        n = len(points["obs"])
        z3, z1 = np.zeros((n, 3), np.float32), np.zeros(n, np.float32)
        a = dict(pos=np.ascontiguousarray(points["pos"], np.float32), normal=np.ascontiguousarray(points.get("normal", z3), np.float32),
                 mn=np.ascontiguousarray(points.get("min_dist", z1), np.float32), mx=np.ascontiguousarray(points.get("max_dist", z1), np.float32),
                 desc=np.ascontiguousarray(points["desc"], np.uint8), obs=np.ascontiguousarray(points["obs"], np.int32),
                 bad=np.ascontiguousarray(points["bad"], np.uint8), local=np.ascontiguousarray(points.get("local", np.zeros(n, np.uint8)), np.uint8))
        return n, a, RumiTrackPoints(n, *(capi.ptr(a[k]) for k in ("pos", "normal", "mn", "mx", "desc", "obs", "bad", "local")), *_stale(points, a))

    def set_distortion(self, K4, dist5):
        """rumi_track_set_distortion: mK and mDistCoef (k1, k2, p1, p2, k3); dist5 None / k1 == 0 switches undistortion off."""
        K = np.ascontiguousarray(K4, np.float32)
        d = np.ascontiguousarray(dist5, np.float32) if dist5 is not None else None
        capi.check(self._lib.rumi_track_set_distortion(self._h, capi.ptr(K), capi.ptr(d) if d is not None else None))

    def undistorted(self):
        """(mvKeysUn of the resident frame, (mnMinX, mnMinY, mnMaxX, mnMaxY))."""
        keys = np.zeros(self.cap, KP_DTYPE); b = np.zeros(4, np.float32)
        capi.check(self._lib.rumi_track_undistorted(self._h, capi.ptr(keys), self.cap, capi.ptr(b)))
        return keys[:self._n].copy(), b

    def last_projections(self, n_points):
        """rumi_track_last_projections: [n_points, 5] f32 = mTrackProjX, mTrackProjY, mnTrackScaleLevel, mTrackViewCos, mTrackDepth of the last SearchLocalPoints."""
        out = np.zeros((int(n_points), 5), np.float32)
        capi.check(self._lib.rumi_track_last_projections(self._h, int(n_points), capi.ptr(out)))
        return out

    @staticmethod
    def _result(res, fields):
        out = {f: getattr(res, f) for f in fields if f not in ("Tcw_motion", "Tcw", "Rcw", "tcw", "Ow")}
        for f in ("Tcw_motion", "Tcw", "Rcw", "tcw", "Ow"):
            if f in fields:
                out[f] = np.array(getattr(res, f), np.float32)
        return out

    def extract(self, img):
        """rumi_track_extract: Frame::ExtractORB; returns (monoIndex, keys, desc) and keeps the frame on the device."""
        img = np.asarray(img, np.uint8)
        assert img.ndim == 2 and img.strides[1] == 1
        h, w = img.shape
        keys = np.zeros(self.cap, KP_DTYPE); desc = np.zeros((self.cap, 32), np.uint8)
        n, mono = C.c_int32(), C.c_int32()
        capi.check(self._lib.rumi_track_extract(self._h, capi.ptr(img), w, h, img.strides[0], capi.ptr(keys), capi.ptr(desc), self.cap, C.byref(n), C.byref(mono)))
        self._n = n.value
        return mono.value, keys[:n.value].copy(), desc[:n.value].copy()

    def motion(self, K4, Tcw_pred7, last_keys, last_mp, last_outlier, points, th_motion=15.0):
        """rumi_track_motion: Tracking::TrackWithMotionModel on the resident frame."""
        K4 = np.ascontiguousarray(K4, np.float32); T = np.ascontiguousarray(Tcw_pred7, np.float32)
        lk = np.ascontiguousarray(last_keys, KP_DTYPE); lm = np.ascontiguousarray(last_mp, np.int32); lo = np.ascontiguousarray(last_outlier, np.uint8)
        _, keep, P = self._points(points, False)
        assert len(lm) >= len(lk) and len(lo) >= len(lk), "last_mp / last_outlier must cover every last-frame key-point (the C entry reads nlast of each)"
        mp = np.full(self.cap, -1, np.int32); dis = np.full(self.cap, -1, np.int32)
        res = RumiTrackResult()
        capi.check(self._lib.rumi_track_motion(self._h, capi.ptr(K4), capi.ptr(T), capi.ptr(lk), len(lk), capi.ptr(lm), capi.ptr(lo), C.byref(P), float(th_motion),
                                               capi.ptr(mp), capi.ptr(dis), C.byref(res)))
        out = self._result(res, ("n", "mono_index", "th_motion", "nmatches_motion", "ngood_motion", "nmatches_map", "Tcw_motion"))
        out.update(frame_mp=mp[:res.n].copy(), discarded=dis[:res.n].copy())
        return out

    def reference_keyframe(self, voc, K4, Tcw_init7, kf_view, kf_fv, kf_mp, points, levelsup=4, nnratio=0.7, check_orientation=True):
        """rumi_track_reference_keyframe: Tracking::TrackReferenceKeyFrame on the resident frame.  voc: rumi_slam_amd.vocabulary.Vocabulary;
        kf_view: matcher.FrameView of the key-frame; kf_fv: matcher.FeatureVectorView (CSR); kf_mp: indices into `points`."""
        K4 = np.ascontiguousarray(K4, np.float32); T = np.ascontiguousarray(Tcw_init7, np.float32)
        km = np.ascontiguousarray(kf_mp, np.int32)
        _, keep, P = self._points(points, False)
        mp = np.full(self.cap, -1, np.int32); dis = np.full(self.cap, -1, np.int32)
        word = np.zeros(self.cap, np.uint32); node = np.zeros(self.cap, np.uint32); wgt = np.zeros(self.cap, np.float64)
        res = RumiTrackResult()
        t0 = time.perf_counter()
        rc = self._lib.rumi_track_reference_keyframe(self._h, voc._h, int(levelsup), capi.ptr(K4), capi.ptr(T), C.byref(kf_view.c), C.byref(kf_fv.c),
                                                     capi.ptr(km), C.byref(P), float(nnratio), int(check_orientation), capi.ptr(word), capi.ptr(wgt),
                                                     capi.ptr(node), capi.ptr(mp), capi.ptr(dis), C.byref(res))
        self.last_call_s = time.perf_counter() - t0                    # the C entry alone (probes)
        capi.check(rc)
        out = self._result(res, ("n", "mono_index", "nmatches_motion", "ngood_motion", "nmatches_map", "Tcw_motion"))
        k = res.n
        out.update(frame_mp=mp[:k].copy(), discarded=dis[:k].copy(), word_id=word[:k].copy(), word_weight=wgt[:k].copy(), node_id=node[:k].copy())
        return out

    def local(self, K4, Tcw7, frame_mp_in, points, seen_in=None, th_local=1.0, far_points=False, th_far_points=50.0):
        """rumi_track_local: Tracking::TrackLocalMap after UpdateLocalMap on the resident frame."""
        K4 = np.ascontiguousarray(K4, np.float32); T = np.ascontiguousarray(Tcw7, np.float32)
        fin = np.ascontiguousarray(frame_mp_in, np.int32)
        n, keep, P = self._points(points, True)
        si = np.ascontiguousarray(seen_in, np.uint8) if seen_in is not None else None
        # the C entry reads frame_mp_in[0 .. n of the resident frame) and seen_in[0 .. points): shorter arrays would be host out-of-bounds reads
        assert getattr(self, "_n", None) is None or len(fin) >= self._n, f"frame_mp_in has {len(fin)} entries, the resident frame {self._n} features"
        assert si is None or len(si) >= n, f"seen_in has {len(si)} entries for {n} points"
        mp = np.full(self.cap, -1, np.int32); outl = np.zeros(self.cap, np.uint8); in_view = np.zeros(max(n, 1), np.uint8)
        res = RumiTrackResult()
        capi.check(self._lib.rumi_track_local(self._h, capi.ptr(K4), capi.ptr(T), capi.ptr(fin), C.byref(P), capi.ptr(si) if si is not None else None,
                                              float(th_local), int(far_points), float(th_far_points), capi.ptr(mp), capi.ptr(outl), capi.ptr(in_view), C.byref(res)))
        out = self._result(res, ("n", "n_to_match", "nmatches_local", "ngood_local", "matches_inliers", "Tcw", "Rcw", "tcw", "Ow"))
        out.update(frame_mp=mp[:res.n].copy(), outlier=outl[:res.n].copy(), in_view=in_view[:n].copy())
        return out
