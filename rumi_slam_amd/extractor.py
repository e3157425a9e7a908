"""Host-side mirror of ``ORB_SLAM3::ORBextractor`` (R/include/cloud_edge_slam_lib/ORBextractor.h:42-111)
over the C ABI.  Same constructor arguments, same call semantics (``__call__`` returns the
reference's ``monoIndex`` and fills key-points / descriptors), same getters.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import KP_DTYPE, RumiOrbConfig


class ORBextractor:
    HARRIS_SCORE, FAST_SCORE = 0, 1

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, max_width=640, max_height=480,
                 max_batch=1, device=-1, blur_variant=0):
        self._lib = capi.lib()
        self._resident = False
        self.cfg = RumiOrbConfig(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, max_width, max_height,
                                 max_batch, device, 0, blur_variant)
        self._h = C.c_void_p()
        capi.check(self._lib.rumi_orb_create(C.byref(self.cfg), C.byref(self._h)))
        self.nfeatures, self.nlevels = nfeatures, nlevels
        self._tables = tables(nfeatures, scaleFactor, nlevels)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.rumi_orb_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    # ---- getters of the reference class ----
    def GetLevels(self):
        return self.nlevels

    def GetScaleFactor(self):
        return float(np.float32(self.cfg.scale_factor))

    def GetScaleFactors(self):
        return self._tables["scale"].copy()

    def GetInverseScaleFactors(self):
        return self._tables["inv_scale"].copy()

    def GetScaleSigmaSquares(self):
        return self._tables["sigma2"].copy()

    def GetInverseScaleSigmaSquares(self):
        return self._tables["inv_sigma2"].copy()

    # ---- operator() ----
    def image_buffer(self, w, h):
        """rumi_orb_image_buffer: an [h, w] uint8 view of the handle's pinned staging memory; a frame written there and passed to __call__ is
        uploaded without the staging copy."""
        buf, stride = C.c_void_p(), C.c_int32()
        capi.check(self._lib.rumi_orb_image_buffer(self._h, int(w), int(h), C.byref(buf), C.byref(stride)))
        flat = np.ctypeslib.as_array((C.c_uint8 * (stride.value * h)).from_address(buf.value))
        return flat.reshape(h, stride.value)[:, :w]

    def __call__(self, image, mask=None, vLappingArea=(0, 1000)):
        """Returns (monoIndex, keypoints[KP_DTYPE], descriptors[n,32] u8); monoIndex == -1 for an empty image."""
        if image is None or image.size == 0:
            return -1, np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        assert image.dtype == np.uint8 and image.ndim == 2, "CV_8UC1 expected"
        if image.strides[1] != 1:                                   # rows must be dense; any row pitch goes through as `stride`
            image = np.ascontiguousarray(image)
        h, w = image.shape
        cap = self.nfeatures + 4 * self.nlevels + 64
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n, mono = C.c_int32(), C.c_int32()
        capi.check(self._lib.rumi_orb_extract(self._h, capi.ptr(image), w, h, image.strides[0], int(vLappingArea[0]),
                                              int(vLappingArea[1]), capi.ptr(kps), capi.ptr(desc), cap,
                                              C.byref(n), C.byref(mono)))
        return mono.value, kps[:n.value].copy(), desc[:n.value].copy()

    # ---- batched, device-resident form (torch tensors on the handle's GPU) ----
    def extract_batch(self, frames, vLappingArea=(0, 1000), cap=None, stream=None, wait=True, out=None):
        """frames: torch.uint8 CUDA tensor [B,H,W] (dense rows, any row pitch).  Returns (kp [B,cap,7] f32 view of the
        28-byte records, desc [B,cap,32] u8, counts [B,2] i32 = (n, monoIndex)) as CUDA tensors.
        wait=False only enqueues (rumi_orb_extract_batch_device_async): the outputs are valid in stream order, device-side
        conditions are reported by the next ``sync()``; `frames` must stay alive until then.
        out=(kp, desc, counts): tensors of those shapes to write into -- REQUIRED with a resident queue, where nothing orders the kernels behind
        the caller's stream: buffers allocated (let alone zeroed) on that stream at call time are not safe to write to."""
        import torch
        assert frames.is_cuda and frames.dtype == torch.uint8 and frames.dim() == 3 and frames.stride(2) == 1
        B, H, W = frames.shape
        cap = cap or (self.nfeatures + 4 * self.nlevels + 64)
        if out is not None:
            kp, desc, counts = out
            assert kp.shape == (B, cap, 7) and kp.dtype == torch.float32 and kp.is_contiguous() and desc.shape == (B, cap, 32) and desc.dtype == torch.uint8
            assert desc.is_contiguous() and counts.shape == (B, 2) and counts.dtype == torch.int32 and counts.is_contiguous()
        else:
            if self._resident:
                raise ValueError("resident queue: pass preallocated out=(kp, desc, counts) buffers (see the docstring)")
            kp = torch.empty((B, cap, 7), dtype=torch.float32, device=frames.device)
            desc = torch.empty((B, cap, 32), dtype=torch.uint8, device=frames.device)
            counts = torch.zeros((B, 2), dtype=torch.int32, device=frames.device)
        st = stream if stream is not None else torch.cuda.current_stream(frames.device)
        fn = self._lib.rumi_orb_extract_batch_device if wait else self._lib.rumi_orb_extract_batch_device_async
        capi.check(fn(
            self._h, frames.data_ptr(), B, W, H, frames.stride(1), frames.stride(0), int(vLappingArea[0]),
            int(vLappingArea[1]), kp.data_ptr(), desc.data_ptr(), counts.data_ptr(), cap, st.cuda_stream))
        return kp, desc, counts

    def extract_batch_records(self, frames, vLappingArea=(0, 1000), cap=None, stream=None, wait=True, out=None):
        """Like ``extract_batch`` with ONE fixed-capacity record per frame as output (``rumination.record_bytes(cap)`` bytes each:
        n, monoIndex, key-points, descriptors): a torch u8 CUDA tensor [B, record_bytes], the payload of the queue's single all-gather.
        ``rumination.record_views`` gives (kp, desc, counts) views of it.  `out`: a tensor to write into (e.g. a padded block)."""
        import torch
        from . import rumination
        assert frames.is_cuda and frames.dtype == torch.uint8 and frames.dim() == 3 and frames.stride(2) == 1
        B, H, W = frames.shape
        cap = cap or (self.nfeatures + 4 * self.nlevels + 64)
        rb = rumination.record_bytes(cap)
        if out is None and self._resident:
            raise ValueError("resident queue: pass a preallocated out= record buffer")
        rec = out if out is not None else torch.zeros((B, rb), dtype=torch.uint8, device=frames.device)
        assert rec.is_cuda and rec.dtype == torch.uint8 and rec.shape[0] >= B and rec.shape[1] == rb and rec.is_contiguous()
        st = stream if stream is not None else torch.cuda.current_stream(frames.device)
        capi.check(self._lib.rumi_orb_extract_batch_records_async(
            self._h, frames.data_ptr(), B, W, H, frames.stride(1), frames.stride(0), int(vLappingArea[0]), int(vLappingArea[1]),
            rec.data_ptr(), rb, cap, st.cuda_stream))
        if wait:
            self.sync()
        return rec

    def extract_batch_host(self, frames, vLappingArea=(0, 1000), cap=None, stream=None, to_host=False):
        """The rumination queue as the reference holds it: `frames` is a list of HOST images (numpy u8 [H,W], dense rows, one shape) or one
        host array / CPU tensor [B,H,W] (pinned memory is copied in place).  Transfers overlap the extraction (rumi_orb_extract_batch_host).
        Returns the device tensors of ``extract_batch``; with to_host=True (pageable arrays) or "pinned" also numpy copies (kp, desc, counts) as a second tuple."""
        import torch
        if hasattr(frames, "numpy") and not isinstance(frames, np.ndarray):      # CPU torch tensor (possibly pinned)
            assert not frames.is_cuda and frames.dtype == torch.uint8 and frames.dim() == 3 and frames.stride(2) == 1
            B, H, W = frames.shape
            ptrs = [frames.data_ptr() + f * frames.stride(0) for f in range(B)]
            pitch, keep = frames.stride(1), frames
        else:
            lst = list(frames)
            keep = [a if a.strides[1] == 1 else np.ascontiguousarray(a) for a in lst]
            B, (H, W) = len(keep), keep[0].shape
            assert all(a.dtype == np.uint8 and a.shape == (H, W) and a.strides[0] == keep[0].strides[0] for a in keep)
            ptrs, pitch = [a.ctypes.data for a in keep], keep[0].strides[0]
        arr = (C.c_void_p * B)(*ptrs)
        dev = torch.device("cuda", torch.cuda.current_device())
        cap = cap or (self.nfeatures + 4 * self.nlevels + 64)
        kp = torch.empty((B, cap, 7), dtype=torch.float32, device=dev)
        desc = torch.empty((B, cap, 32), dtype=torch.uint8, device=dev)
        counts = torch.zeros((B, 2), dtype=torch.int32, device=dev)
        hk = hd = hc = None
        if to_host == "pinned":             # pinned host arrays (with RUMI_ORB_MIRROR=2 every sub-chunk's rows come back behind its kernels; measured slower)
            hk = torch.zeros((B, cap, 28), dtype=torch.uint8).pin_memory().numpy().view(KP_DTYPE).reshape(B, cap)
            hd, hc = torch.zeros((B, cap, 32), dtype=torch.uint8).pin_memory().numpy(), torch.zeros((B, 2), dtype=torch.int32).pin_memory().numpy()
        elif to_host:
            hk, hd, hc = np.zeros((B, cap), KP_DTYPE), np.zeros((B, cap, 32), np.uint8), np.zeros((B, 2), np.int32)
        st = stream if stream is not None else torch.cuda.current_stream(dev)
        capi.check(self._lib.rumi_orb_extract_batch_host(
            self._h, arr, B, W, H, pitch, int(vLappingArea[0]), int(vLappingArea[1]), kp.data_ptr(), desc.data_ptr(), counts.data_ptr(), cap,
            capi.ptr(hk) if to_host else None, capi.ptr(hd) if to_host else None, capi.ptr(hc) if to_host else None, st.cuda_stream))
        del keep
        return ((kp, desc, counts), (hk, hd, hc)) if to_host else (kp, desc, counts)

    def sync(self):
        """Waits for every batch enqueued with wait=False and raises on a device-side condition (rumi_orb_sync)."""
        capi.check(self._lib.rumi_orb_sync(self._h))

    # ---- taps ----
    def pyramid_level(self, level, frame=0, blurred=False, border=0):
        w, h = C.c_int32(), C.c_int32()
        capi.check(self._lib.rumi_orb_pyramid_level(self._h, frame, level, int(blurred), border, None, 0,
                                                    C.byref(w), C.byref(h)))
        out = np.zeros((h.value + 2 * border, w.value + 2 * border), np.uint8)
        capi.check(self._lib.rumi_orb_pyramid_level(self._h, frame, level, int(blurred), border, capi.ptr(out),
                                                    out.strides[0], C.byref(w), C.byref(h)))
        return out

    def stage_keypoints(self, level, stage, frame=0):
        n = C.c_int32()
        capi.check(self._lib.rumi_orb_stage_keypoints(self._h, frame, level, stage, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), KP_DTYPE)
        capi.check(self._lib.rumi_orb_stage_keypoints(self._h, frame, level, stage, capi.ptr(out), len(out), C.byref(n)))
        return out[:n.value]

    def set_resident_queue(self, on=True):
        """The frames of the following batched calls do not depend on work pending on the caller's stream (a queue that sits in device memory):
        back-to-back asynchronous calls then overlap like the sub-chunks of one large call (include/rumi_orb.h).  on: True = four slots
        (calls / sub-chunks in flight), an integer 2 .. 8 = that many."""
        capi.check(self._lib.rumi_orb_set_resident_queue(self._h, int(on)))
        self._resident = bool(on)

    def wait_event(self, event):
        """The next batched call starts behind `event` (a recorded torch.cuda.Event): the end of whoever still reads the buffers the call is
        about to overwrite (rumi_orb_wait_event)."""
        self._wait_event_ref = event          # the library keeps the raw hipEvent_t until the next batched call: the torch Event must outlive it
        capi.check(self._lib.rumi_orb_wait_event(self._h, event.cuda_event if event is not None else None))

    def set_profiling(self, on=True):
        capi.check(self._lib.rumi_orb_set_profiling(self._h, int(on)))

    def stage_ms(self):
        ms = np.zeros(8, np.float32)
        capi.check(self._lib.rumi_orb_stage_ms(self._h, capi.ptr(ms)))
        return dict(pyramid=ms[0], fast=ms[1], compact=ms[2], blur=ms[3], quadtree=ms[4], orient_desc=ms[5],
                    total=ms[6])


def tables(nfeatures=1000, scaleFactor=1.2, nlevels=8):
    """Constructor tables of the reference (host-only; works without a GPU)."""
    cfg = RumiOrbConfig(nfeatures, scaleFactor, nlevels, 20, 7, 640, 480, 1, -1, 0, 0)
    f = [np.zeros(nlevels, np.float32) for _ in range(4)]
    per = np.zeros(nlevels, np.int32)
    umax = np.zeros(16, np.int32)
    capi.check(capi.lib().rumi_orb_tables(C.byref(cfg), *[capi.ptr(a) for a in f], capi.ptr(per), capi.ptr(umax)))
    return dict(scale=f[0], inv_scale=f[1], sigma2=f[2], inv_sigma2=f[3], per_level=per, umax=umax)
