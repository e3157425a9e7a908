"""Mirror of DBoW2's ORBVocabulary::transform over the C ABI of include/rumi_voc.h (ctypes; host logic only)."""
import ctypes as C

import numpy as np

from . import capi

VOC_SYMBOLS = ["rumi_voc_create", "rumi_voc_load_text", "rumi_voc_destroy", "rumi_voc_words", "rumi_voc_levels", "rumi_voc_set_levels", "rumi_voc_transform_features",
               "rumi_voc_transform_batch_device", "rumi_voc_transform", "rumi_voc_assemble"]
TF_IDF, TF, IDF, BINARY = 0, 1, 2, 3
L1_NORM, L2_NORM, CHI_SQUARE, KL, BHATTACHARYYA, DOT_PRODUCT = range(6)


def _lib():
    L = capi.lib()
    if getattr(L, "_voc_ready", False):
        return L
    vp, i32 = C.c_void_p, C.c_int32
    L.rumi_voc_create.argtypes = [i32, vp, vp, vp, vp, i32, i32, i32, C.POINTER(vp)]
    L.rumi_voc_load_text.argtypes = [C.c_char_p, i32, C.POINTER(vp)]
    L.rumi_voc_destroy.argtypes = [vp]
    L.rumi_voc_destroy.restype = None
    L.rumi_voc_words.argtypes = [vp]
    L.rumi_voc_levels.argtypes = [vp]
    L.rumi_voc_set_levels.argtypes = [vp, i32]
    L.rumi_voc_transform_features.argtypes = [vp, vp, i32, i32, vp, vp, vp]
    L.rumi_voc_transform_batch_device.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]
    L.rumi_voc_transform.argtypes = [vp, vp, i32, i32, vp, vp, C.POINTER(i32), vp, vp, vp, C.POINTER(i32)]
    L.rumi_voc_assemble.argtypes = [vp, i32, vp, vp, vp, vp, vp, C.POINTER(i32), vp, vp, vp, C.POINTER(i32)]
    L._voc_ready = True
    return L


class ORBVocabulary:
    """Vocabulary tree on the GPU.  Build from the node table of the text format (parent, is_leaf, descriptor, weight per node,
    node 0 = root) or from an ORBvoc.txt-style file."""

    def __init__(self, parent=None, is_leaf=None, desc=None, weight=None, weighting=TF_IDF, scoring=L1_NORM, path=None, device=-1):
        self._lib = _lib()
        self._h = C.c_void_p()
        if path is not None:
            capi.check(self._lib.rumi_voc_load_text(str(path).encode(), device, C.byref(self._h)))
        else:
            p = np.ascontiguousarray(parent, np.int32); l = np.ascontiguousarray(is_leaf, np.uint8)
            d = np.ascontiguousarray(desc, np.uint8); w = np.ascontiguousarray(weight, np.float64)
            capi.check(self._lib.rumi_voc_create(len(p), capi.ptr(p), capi.ptr(l), capi.ptr(d), capi.ptr(w), int(weighting), int(scoring), device,
                                                 C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.rumi_voc_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def size(self):
        return self._lib.rumi_voc_words(self._h)

    def levels(self):
        return self._lib.rumi_voc_levels(self._h)

    def transform_features(self, desc, levelsup=4):
        d = np.ascontiguousarray(desc, np.uint8)
        n = len(d)
        word, node, w = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.float64)
        capi.check(self._lib.rumi_voc_transform_features(self._h, capi.ptr(d), n, int(levelsup), capi.ptr(word), capi.ptr(w), capi.ptr(node)))
        return word, w, node

    def transform(self, desc, levelsup=4):
        """Returns (BowVector as (word ids, values), FeatureVector as (node ids, offsets, indices))."""
        d = np.ascontiguousarray(desc, np.uint8)
        n = len(d)
        bi, bv = np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.float64)
        fn, fo, fi = np.zeros(max(n, 1), np.uint32), np.zeros(n + 1, np.int32), np.zeros(max(n, 1), np.uint32)
        nw, nn = C.c_int32(), C.c_int32()
        capi.check(self._lib.rumi_voc_transform(self._h, capi.ptr(d), n, int(levelsup), capi.ptr(bi), capi.ptr(bv), C.byref(nw), capi.ptr(fn), capi.ptr(fo),
                                                capi.ptr(fi), C.byref(nn)))
        return (bi[:nw.value].copy(), bv[:nw.value].copy()), (fn[:nn.value].copy(), fo[:nn.value + 1].copy(), fi[:fo[nn.value]].copy())

    def transform_batch(self, desc, counts, levelsup=4, stream=None):
        """desc [B,cap,32] u8 and counts [B,2] i32 CUDA tensors (the extractor's batch outputs) -> (word [B,cap] i32-viewed u32,
        weight [B,cap] f64, node [B,cap]) CUDA tensors; slots >= n are left as allocated (zeros)."""
        import torch
        B, cap, _ = desc.shape
        word = torch.zeros((B, cap), dtype=torch.int32, device=desc.device)
        node = torch.zeros((B, cap), dtype=torch.int32, device=desc.device)
        w = torch.zeros((B, cap), dtype=torch.float64, device=desc.device)
        st = stream if stream is not None else torch.cuda.current_stream(desc.device).cuda_stream
        capi.check(self._lib.rumi_voc_transform_batch_device(self._h, desc.data_ptr(), counts.data_ptr(), B, cap, int(levelsup), word.data_ptr(),
                                                             w.data_ptr(), node.data_ptr(), st))
        return word, w, node

    def assemble(self, word_id, weight, node_id):
        """BowVector and FeatureVector from the per-feature transform (rumi_voc_assemble): ((bow_ids, bow_vals), (fv_nodes, fv_offsets, fv_indices))."""
        w = np.ascontiguousarray(word_id, np.uint32); v = np.ascontiguousarray(weight, np.float64); nd = np.ascontiguousarray(node_id, np.uint32)
        n = len(w)
        bi, bv = np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.float64)
        fn, fo, fi = np.zeros(max(n, 1), np.uint32), np.zeros(n + 1, np.int32), np.zeros(max(n, 1), np.uint32)
        nw, nn = C.c_int32(), C.c_int32()
        capi.check(self._lib.rumi_voc_assemble(self._h, n, capi.ptr(w), capi.ptr(v), capi.ptr(nd), capi.ptr(bi), capi.ptr(bv), C.byref(nw), capi.ptr(fn),
                                               capi.ptr(fo), capi.ptr(fi), C.byref(nn)))
        return (bi[:nw.value].copy(), bv[:nw.value].copy()), (fn[:nn.value].copy(), fo[:nn.value + 1].copy(), fi[:fo[nn.value]].copy())
