"""ctypes binding of librumi_hip.so — the C ABI declared in include/rumi_orb.h.

Raises at import of the library if it has not been built (``python __graft_entry__.py`` or
``make -C rumi_slam_amd/csrc``); there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librumi_hip.so")

RUMI_OK, RUMI_E_INVALID, RUMI_E_NO_DEVICE, RUMI_E_CAPACITY, RUMI_E_EMPTY = 0, -1, -2, -3, -4

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28


class RumiOrbConfig(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32), ("max_width", C.c_int32),
                ("max_height", C.c_int32), ("max_batch", C.c_int32), ("device", C.c_int32),
                ("host_threads", C.c_int32), ("blur_variant", C.c_int32)]


class RumiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rumi status {code}: {msg}")
        self.code = code


_lib = None

# every symbol include/rumi_orb.h declares (tests check the library exports all of them)
ORB_SYMBOLS = ["rumi_last_error", "rumi_device_count", "rumi_orb_create", "rumi_orb_destroy", "rumi_orb_tables",
               "rumi_orb_extract", "rumi_orb_image_buffer", "rumi_orb_extract_batch_device", "rumi_orb_extract_batch_device_async", "rumi_orb_sync", "rumi_orb_set_resident_queue", "rumi_orb_wait_event", "rumi_orb_extract_batch_records_async", "rumi_orb_extract_batch_host", "rumi_orb_extract_batch_host_records",
               "rumi_orb_pyramid_level",
               "rumi_orb_stage_keypoints", "rumi_orb_set_profiling", "rumi_orb_stage_ms"]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build the HIP library first (python __graft_entry__.py). "
                           "There is no CPU fallback for the product path.")
    # PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64; two HSA runtimes in one process
    # cannot both open the GPU.  Load torch's first (when torch is present) so that this library binds to
    # the copy already in the process; a C++ host without torch binds to /opt/rocm as usual.
    if os.environ.get("RUMI_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.rumi_last_error.restype = C.c_char_p
    L.rumi_device_count.restype = C.c_int
    L.rumi_orb_create.argtypes = [C.POINTER(RumiOrbConfig), C.POINTER(vp)]
    L.rumi_orb_destroy.argtypes = [vp]
    L.rumi_orb_destroy.restype = None
    L.rumi_orb_tables.argtypes = [C.POINTER(RumiOrbConfig)] + [vp] * 6
    L.rumi_orb_extract.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, C.POINTER(i32), C.POINTER(i32)]
    L.rumi_orb_image_buffer.argtypes = [vp, i32, i32, C.POINTER(vp), C.POINTER(i32)]
    L.rumi_orb_extract_batch_device.argtypes = [vp, vp, i32, i32, i32, i32, i64, i32, i32, vp, vp, vp, i32, vp]
    L.rumi_orb_extract_batch_device_async.argtypes = L.rumi_orb_extract_batch_device.argtypes
    L.rumi_orb_sync.argtypes = [vp]
    L.rumi_orb_extract_batch_records_async.argtypes = [vp, vp, i32, i32, i32, i32, i64, i32, i32, vp, i64, i32, vp]
    L.rumi_orb_extract_batch_host.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, i32, vp, vp, vp, vp]
    L.rumi_orb_extract_batch_host_records.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp, i64, i32, vp, vp]
    L.rumi_orb_pyramid_level.argtypes = [vp, i32, i32, i32, i32, vp, i32, C.POINTER(i32), C.POINTER(i32)]
    L.rumi_orb_stage_keypoints.argtypes = [vp, i32, i32, i32, vp, i32, C.POINTER(i32)]
    L.rumi_orb_set_profiling.argtypes = [vp, i32]
    L.rumi_orb_set_resident_queue.argtypes = [vp, i32]
    L.rumi_orb_wait_event.argtypes = [vp, vp]
    L.rumi_orb_stage_ms.argtypes = [vp, vp]
    _lib = L
    return L


def check(code):
    if code != RUMI_OK:
        raise RumiError(code, lib().rumi_last_error().decode("utf-8", "replace"))


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


MATCH_SYMBOLS = ["rumi_descriptor_distance", "rumi_match_create", "rumi_match_destroy", "rumi_search_by_projection_mappoints",
                 "rumi_search_by_projection_frame", "rumi_search_by_bow", "rumi_search_by_bow_kf", "rumi_search_by_projection_sim3",
                 "rumi_search_by_projection_reloc", "rumi_search_for_initialization", "rumi_search_for_triangulation", "rumi_fuse_candidates", "rumi_search_by_sim3", "rumi_frame_is_in_frustum", "rumi_search_local_points", "rumi_search_by_bow_batch", "rumi_match_bruteforce_batch_device", "rumi_match_bruteforce_batch_device_strided", "rumi_match_bruteforce_ring_device"]

OPT_SYMBOLS = ["rumi_opt_create", "rumi_opt_destroy", "rumi_pose_optimization", "rumi_pose_optimization_batch", "rumi_local_ba", "rumi_local_ba_batch", "rumi_merge_ba", "rumi_bundle_adjustment", "rumi_sim3_inliers",
               "rumi_optimize_sim3", "rumi_sim3_ransac", "rumi_opt_stage_ms", "rumi_opt_set_profiling", "rumi_opt_kernel_ms"]

VOC_SYMBOLS = ["rumi_voc_create", "rumi_voc_load_text", "rumi_voc_destroy", "rumi_voc_words", "rumi_voc_levels", "rumi_voc_set_levels", "rumi_voc_assemble", "rumi_voc_transform_features",
               "rumi_voc_transform_batch_device", "rumi_voc_transform"]
TRACK_SYMBOLS = ["rumi_track_create", "rumi_track_destroy", "rumi_track_frame", "rumi_track_extract", "rumi_track_motion",
                 "rumi_track_reference_keyframe", "rumi_track_local", "rumi_track_image_buffer", "rumi_track_last_projections", "rumi_track_set_distortion", "rumi_track_undistorted"]
QUEUE_SYMBOLS = ["rumi_queue_create", "rumi_queue_destroy", "rumi_queue_shards", "rumi_queue_record_bytes", "rumi_queue_block_capacity", "rumi_queue_row",
                 "rumi_queue_uses_rccl", "rumi_queue_extract", "rumi_queue_last_ms"]
HOOK_SYMBOLS = ["rumi_hook_sort_like_std", "rumi_hook_sort_device", "rumi_hook_std_sort", "rumi_hook_quadtree", "rumi_hook_sinf", "rumi_hook_cosf",
                "rumi_hook_fast_atan2", "rumi_hook_cv_round", "rumi_hook_magic_div"]


def hooks():
    """Host-only test hooks (include/rumi_testhooks.h)."""
    L = lib()
    if getattr(L, "_hooks_ready", False):
        return L
    vp, i32 = C.c_void_p, C.c_int32
    L.rumi_hook_sort_like_std.argtypes = [vp, vp, i32]
    L.rumi_hook_sort_device.argtypes = [vp, vp, i32]
    L.rumi_hook_std_sort.argtypes = [vp, vp, i32]
    L.rumi_hook_quadtree.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, i32, C.POINTER(i32)]
    for name in ("rumi_hook_sinf", "rumi_hook_cosf"):
        getattr(L, name).restype = C.c_float
        getattr(L, name).argtypes = [C.c_float]
    L.rumi_hook_fast_atan2.restype = C.c_float
    L.rumi_hook_fast_atan2.argtypes = [C.c_float, C.c_float]
    L.rumi_hook_cv_round.argtypes = [C.c_float]
    L.rumi_hook_magic_div.argtypes = [i32, i32]
    L._hooks_ready = True
    return L
