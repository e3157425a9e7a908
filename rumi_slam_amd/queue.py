"""Host-side mirror of include/rumi_queue.h: the rumination queue on the GPUs of one node from ONE process (ctypes; host logic only)."""
import ctypes as C

import numpy as np

from . import capi
from .capi import RumiOrbConfig

QUEUE_SYMBOLS = ["rumi_queue_create", "rumi_queue_destroy", "rumi_queue_shards", "rumi_queue_record_bytes", "rumi_queue_block_capacity", "rumi_queue_row",
                 "rumi_queue_uses_rccl", "rumi_queue_extract", "rumi_queue_last_ms", "rumi_orb_extract_batch_host_records"]


def _lib():
    L = capi.lib()
    if getattr(L, "_queue_ready", False):
        return L
    vp, i32 = C.c_void_p, C.c_int32
    L.rumi_queue_create.argtypes = [vp, vp, i32, i32, C.POINTER(vp)]
    L.rumi_queue_destroy.argtypes = [vp]
    L.rumi_queue_destroy.restype = None
    L.rumi_queue_shards.argtypes = [vp]
    L.rumi_queue_record_bytes.argtypes = [vp]
    L.rumi_queue_record_bytes.restype = C.c_int64
    L.rumi_queue_block_capacity.argtypes = [vp]
    L.rumi_queue_row.argtypes = [vp, i32, i32]
    L.rumi_queue_uses_rccl.argtypes = [vp]
    L.rumi_queue_extract.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]
    L.rumi_queue_last_ms.argtypes = [vp, vp]
    L._queue_ready = True
    return L


class RuminationQueue:
    """rumi_queue_create: `devices` = one HIP ordinal per shard (an ordinal may repeat: logical shards on one device, the exchange is then made of
    device-to-device copies); max_block = the largest block a shard will see."""

    def __init__(self, nfeatures, scale_factor, nlevels, ini_th, min_th, devices, max_block, max_width=640, max_height=480, cap=None, blur_variant=0):
        self._lib = _lib()
        self.cap = int(cap) if cap is not None else nfeatures + 96
        cfg = RumiOrbConfig(nfeatures, scale_factor, nlevels, ini_th, min_th, max_width, max_height, int(max_block), -1, 0, blur_variant)
        dev = np.ascontiguousarray(devices, np.int32)
        self._h = C.c_void_p()
        capi.check(self._lib.rumi_queue_create(C.byref(cfg), capi.ptr(dev), len(dev), self.cap, C.byref(self._h)))
        self.n_shards = len(dev)
        self.record_bytes = int(self._lib.rumi_queue_record_bytes(self._h))
        self.block_capacity = int(self._lib.rumi_queue_block_capacity(self._h))
        self.uses_rccl = bool(self._lib.rumi_queue_uses_rccl(self._h))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.rumi_queue_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def row(self, n_frames, frame):
        return int(self._lib.rumi_queue_row(self._h, int(n_frames), int(frame)))

    def extract(self, frames, lapping=(0, 1000), want_host=True, out=None):
        """frames: list of HxW u8 arrays (or one [F,H,W] array), time order.  Returns (records [F, record_bytes] u8 or None, device pointers of the
        gathered queue per shard)."""
        if isinstance(frames, np.ndarray) and frames.ndim == 3 and frames.dtype == np.uint8 and frames.strides[2] == 1:
            F, H, W = frames.shape                                   # one array: the frame pointers by arithmetic (a list costs ~0.7 us per frame)
            fr = [frames]
            ptrs = frames.ctypes.data + np.arange(F, dtype=np.uint64) * np.uint64(frames.strides[0])
            row_stride = frames.strides[1]
        else:
            fr = [f if (f.dtype == np.uint8 and f.flags.c_contiguous) else np.ascontiguousarray(f, np.uint8) for f in frames]
            F, (H, W) = len(fr), fr[0].shape
            ptrs = np.fromiter((f.__array_interface__["data"][0] for f in fr), np.uint64, F)
            row_stride = fr[0].strides[0]
        dg = (C.c_void_p * self.n_shards)()
        rec = out if out is not None else (np.zeros((F, self.record_bytes), np.uint8) if want_host else None)
        assert rec is None or (rec.shape == (F, self.record_bytes) and rec.dtype == np.uint8 and rec.flags.c_contiguous)
        capi.check(self._lib.rumi_queue_extract(self._h, capi.ptr(ptrs), F, W, H, row_stride, int(lapping[0]), int(lapping[1]),
                                                C.cast(dg, C.c_void_p), capi.ptr(rec) if rec is not None else None))
        return rec, [int(p or 0) for p in dg]

    def last_ms(self):
        ms = np.zeros(4, np.float32)
        capi.check(self._lib.rumi_queue_last_ms(self._h, capi.ptr(ms)))
        return dict(extraction=float(ms[0]), exchange=float(ms[1]), copy_back=float(ms[2]), total=float(ms[3]))


def split_records(records, cap):
    """[F, record_bytes] u8 host records -> (counts [F,2] i32, kp [F,cap] structured 28-byte key-points as raw bytes [F,cap,28], desc [F,cap,32])."""
    F = records.shape[0]
    counts = records[:, :8].copy().view(np.int32).reshape(F, 2)
    kp = records[:, 8:8 + 28 * cap].reshape(F, cap, 28)
    desc = records[:, 8 + 28 * cap:8 + 60 * cap].reshape(F, cap, 32)
    return counts, kp, desc
