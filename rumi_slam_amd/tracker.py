"""Host-side mirror of include/rumi_track.h: one device-resident Tracking step (TrackWithMotionModel + TrackLocalMap data path,
R/lib_src/Tracking.cc:2441-2607, 2996-3055) behind one call."""
import ctypes as C

import numpy as np

from . import capi
from .capi import KP_DTYPE, RumiOrbConfig


class RumiTrackPoints(C.Structure):
    _fields_ = [("n", C.c_int32), ("pos", C.c_void_p), ("normal", C.c_void_p), ("min_dist", C.c_void_p), ("max_dist", C.c_void_p),
                ("desc", C.c_void_p), ("obs", C.c_void_p), ("bad", C.c_void_p), ("local", C.c_void_p)]


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
        P = RumiTrackPoints(n, *(capi.ptr(a[k]) for k in ("pos", "normal", "mn", "mx", "desc", "obs", "bad", "local")))
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
