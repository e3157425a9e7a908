/*
 * rumi_orb.h — C ABI of the MI355X-native ORB front-end (librumi_hip.so).
 *
 * Drop-in boundary for the reference class ORB_SLAM3::ORBextractor
 * (R/ = /root/reference/src/rumi-slam/):
 *   R/include/cloud_edge_slam_lib/ORBextractor.h:42-111   class surface
 *   R/lib_src/ORBextractor.cc:405-461                      constructor tables
 *   R/lib_src/ORBextractor.cc:1014-1091                    operator()
 * The reference has no FFI layer; the C++ facade in rumi-slam_amd/facade/ keeps the reference's
 * class signature and marshals cv::Mat / std::vector<cv::KeyPoint> into these calls
 * (INTEGRATION.md shows the binding).
 *
 * Conventions: every function returns an int status (RUMI_OK = 0, < 0 = error); no exceptions
 * cross the ABI; all buffers are caller-owned; a handle may be used from any host thread but not
 * concurrently (same rule as the reference object, which mutates mvImagePyramid).  There is no CPU
 * fallback: without a gfx950 device every entry point that computes returns RUMI_E_NO_DEVICE.
 */
#ifndef RUMI_ORB_H
#define RUMI_ORB_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RUMI_OK 0
#define RUMI_E_INVALID (-1)     /* bad argument / unsupported geometry */
#define RUMI_E_NO_DEVICE (-2)   /* no HIP device, or a HIP call failed (see rumi_last_error) */
#define RUMI_E_CAPACITY (-3)    /* caller buffer or handle capacity too small */
#define RUMI_E_EMPTY (-4)       /* empty image: the reference's operator() returns -1 here */

/* Same 28-byte layout as cv::KeyPoint: pt.x, pt.y, size, angle, response, octave, class_id. */
typedef struct RumiKeyPoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} RumiKeyPoint;

/* ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
 * (ORBextractor.h:47-48) plus the arena sizes the handle pre-allocates in HBM. */
typedef struct RumiOrbConfig {
    int32_t nfeatures;
    float scale_factor;
    int32_t nlevels;        /* 1..16 */
    int32_t ini_th_fast;
    int32_t min_th_fast;
    int32_t max_width;      /* largest image the handle will see */
    int32_t max_height;
    int32_t max_batch;      /* largest number of frames per batched call (>= 1) */
    int32_t device;         /* HIP device ordinal, -1 = current */
    int32_t host_threads;   /* threads that fill the pinned staging slots of rumi_orb_extract_batch_host (0 = up to 16) */
    int32_t blur_variant;   /* which cv::GaussianBlur(7x7, sigma 2) of 8-bit images to reproduce -- the reference only says "OpenCV 3.4" (R/CMakeLists.txt:35),
                             * and the two differ by +-1 grey level, which can flip rBRIEF bits:
                             * 0 = RUMI_BLUR_FIXED_POINT  OpenCV >= 3.4.2 / 4.x bit-exact fixed-point path: taps {18,34,48,56,48,34,18} / 256 (sum 256)
                             * 1 = RUMI_BLUR_SEPFILTER    OpenCV 3.4.0 / 3.4.1 sepFilter2D path: the float kernel scaled by 256 and rounded tap by tap,
                             *                            {18,34,49,55,49,34,18} / 256 (sum 257), result saturated */
} RumiOrbConfig;
enum { RUMI_BLUR_FIXED_POINT = 0, RUMI_BLUR_SEPFILTER = 1 };

typedef struct RumiOrb RumiOrb;

const char *rumi_last_error(void);          /* thread-local message of the last failing call */
int rumi_device_count(void);                /* HIP devices visible to the process (0 if none) */

/* ORBextractor::ORBextractor — ORBextractor.cc:405-461 */
int rumi_orb_create(const RumiOrbConfig *cfg, RumiOrb **out);
void rumi_orb_destroy(RumiOrb *h);

/* GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares / GetInverseScaleSigmaSquares
 * (ORBextractor.h:62-84) plus mnFeaturesPerLevel and umax; each array has nlevels entries
 * (umax16: 16).  Any pointer may be NULL.  Host-only: works without a device. */
int rumi_orb_tables(const RumiOrbConfig *cfg, float *scale, float *inv_scale, float *sigma2,
                    float *inv_sigma2, int32_t *features_per_level, int32_t *umax16);

/* ORBextractor::operator()(image, mask, keypoints, descriptors, vLappingArea) — ORBextractor.cc:1014-1091.
 * Host image in (8-bit grey, `stride` bytes per row), host key-points / descriptors out.
 * *n_out = number of key-points, *mono_out = the reference's return value (monoIndex).
 * Returns RUMI_E_EMPTY (and *mono_out = -1) for an empty image, RUMI_E_CAPACITY if n > cap
 * (*n_out still holds n). */
int rumi_orb_extract(RumiOrb *h, const uint8_t *img, int32_t w, int32_t hgt, int32_t stride,
                     int32_t lap0, int32_t lap1, RumiKeyPoint *kp_out, uint8_t *desc_out, int32_t cap,
                     int32_t *n_out, int32_t *mono_out);

/* The handle's pinned staging memory for a w x hgt frame (*stride = w rounded up to 4 bytes per row).  Optional: a caller whose camera driver or
 * decoder writes the frame straight into it (the cv::Mat ORBextractor::operator() receives, constructed on this memory) and passes this pointer
 * and stride to rumi_orb_extract saves its staging copy (~20 us for 640 x 480). */
int rumi_orb_image_buffer(RumiOrb *h, int32_t w, int32_t hgt, uint8_t **buf, int32_t *stride);

/* Batched form for the rumination queue (CloudImageSampler.cc:148-170 collects the frames; KFDSample.cc:113
 * runs the same extractor on them).  All pointers are DEVICE pointers; frames are `frame_stride`
 * bytes apart.  Outputs: d_kp [nframes][cap], d_desc [nframes][cap][32], d_counts [nframes][2] =
 * {n, monoIndex}; slots >= n are left untouched.  Work is enqueued on `hip_stream` (a hipStream_t, NULL =
 * default stream).  Frames whose base address, `stride` or `frame_stride` is not a multiple of 4 are first copied
 * into an aligned staging arena; aligned frames are read in place (and must stay alive until the work has run).
 * This form BLOCKS until the work has run and returns its status. */
int rumi_orb_extract_batch_device(RumiOrb *h, const void *d_imgs, int32_t nframes, int32_t w, int32_t hgt,
                                  int32_t stride, int64_t frame_stride, int32_t lap0, int32_t lap1,
                                  void *d_kp, void *d_desc, void *d_counts, int32_t cap, void *hip_stream);
/* The same without waiting: returns once everything is enqueued (argument errors are still reported at once), so the
 * caller can queue the next batch, or other work, behind it.  Device-side conditions (candidate / selection capacity,
 * quadtree limits) accumulate until rumi_orb_sync, which waits for every call enqueued since the last one and returns
 * the first such condition.  Consecutive un-waited calls on one handle must use the same stream (another stream, the
 * profiled path, the taps below and rumi_orb_destroy wait first by themselves). */
int rumi_orb_extract_batch_device_async(RumiOrb *h, const void *d_imgs, int32_t nframes, int32_t w, int32_t hgt,
                                        int32_t stride, int64_t frame_stride, int32_t lap0, int32_t lap1,
                                        void *d_kp, void *d_desc, void *d_counts, int32_t cap, void *hip_stream);
int rumi_orb_sync(RumiOrb *h);
/* Declares the frames of the following batched calls RESIDENT: they do not depend on work pending on `hip_stream` when a call is made (the
 * rumination queue sits in device memory long before it is processed, CloudImageSampler.cc:148-170).  Sub-chunks then never wait for the
 * caller's stream, so back-to-back asynchronous calls overlap like the sub-chunks of one large call -- what a rank's share of a sharded
 * queue (64-128 frames per call) needs to run at the rate of a long one.  The caller's stream still waits for each call's results.
 * The output buffers must be free too (no initialisation queued on `hip_stream`, no reader of their previous contents still running:
 * see rumi_orb_wait_event).  on = 1: four slots (sub-chunks / calls in flight), 2 .. 8: that many (short calls of 64-128 frames want
 * more of them in flight: each is one dependent chain of launches); the handle grows its device arenas to that many slots of
 * min(256, max_batch) frames (a call is cut into sub-chunks of at most 256 frames).  With a resident queue the
 * arenas keep the pyramid of a call's LAST sub-chunk only (rumi_orb_pyramid_level refuses other frames).  Off by default; switching waits
 * for pending calls. */
int rumi_orb_set_resident_queue(RumiOrb *h, int32_t on);
/* With a resident queue nothing orders a call's kernels behind the caller's stream, so what the call reads AND what it overwrites must be
 * free when it is made.  When an output buffer is being reused, hand over the event (a hipEvent_t) that marks its previous consumer's end:
 * the sub-chunks of the NEXT batched call start behind it.  One event per call; NULL clears.  (Without a resident queue the caller's stream
 * waits for the event instead.) */
int rumi_orb_wait_event(RumiOrb *h, void *hip_event);

/* The same work with ONE fixed-capacity record per frame as output,
 *   { int32 n; int32 monoIndex; RumiKeyPoint kp[cap]; uint8 desc[cap][32]; }  = 8 + 60 * cap bytes (record_bytes >= that, multiple of 4),
 * so that the exchange step of the sharded rumination queue is a single all-gather of d_records.  Asynchronous like
 * rumi_orb_extract_batch_device_async. */
int rumi_orb_extract_batch_records_async(RumiOrb *h, const void *d_imgs, int32_t nframes, int32_t w, int32_t hgt,
                                         int32_t stride, int64_t frame_stride, int32_t lap0, int32_t lap1,
                                         void *d_records, int64_t record_bytes, int32_t cap, void *hip_stream);

/* The queue as the reference holds it: `imgs[f]` are HOST frames (CloudImageSampler.cc:148-170 keeps cv::Mats), `stride` bytes per row.
 * Frames go to the device in groups of 64 on a copy stream; a group's extraction waits only for its own transfer, so transfers and
 * kernels overlap.  Pinned frames (hipHostMalloc / hipHostRegister) are copied in place, pageable ones through pinned staging slots filled
 * by cfg.host_threads threads.  Results are written to the DEVICE arrays d_kp / d_desc / d_counts as by rumi_orb_extract_batch_device (the
 * all-gather payload) and, when h_kp / h_desc / h_counts are not NULL, copied to those host arrays of the same shape too.  Blocks until
 * everything has arrived. */
int rumi_orb_extract_batch_host(RumiOrb *h, const uint8_t *const *imgs, int32_t nframes, int32_t w, int32_t hgt, int32_t stride,
                                int32_t lap0, int32_t lap1, void *d_kp, void *d_desc, void *d_counts, int32_t cap,
                                RumiKeyPoint *h_kp, uint8_t *h_desc, int32_t *h_counts, void *hip_stream);

/* The same with ONE record per frame as output (the layout of rumi_orb_extract_batch_records_async): d_records [nframes][record_bytes] on the device,
 * h_records (may be NULL) the same bytes on the host: a pinned h_records receives every sub-chunk's records behind that sub-chunk's kernels, under the
 * kernels of the next ones; a pageable one gets one copy at the end.  What a shard of the rumination queue runs (rumi_queue.h). */
int rumi_orb_extract_batch_host_records(RumiOrb *h, const uint8_t *const *imgs, int32_t nframes, int32_t w, int32_t hgt, int32_t stride,
                                        int32_t lap0, int32_t lap1, void *d_records, int64_t record_bytes, int32_t cap, uint8_t *h_records,
                                        void *hip_stream);

/* Backs the public member mvImagePyramid (ORBextractor.h:86): copies level `level` of frame `frame`
 * of the last call to host memory, with `border` replicated pixels of BORDER_REFLECT_101 on each side
 * (the reference uses 19); which = 0 pyramid, 1 Gaussian-blurred working image (ORBextractor.cc:1057-1058).
 * out may be NULL to query *w_out / *h_out (sizes without border). */
int rumi_orb_pyramid_level(RumiOrb *h, int32_t frame, int32_t level, int32_t which, int32_t border,
                           uint8_t *out, int32_t out_stride, int32_t *w_out, int32_t *h_out);

/* Stage taps for parity tests (results of the last call):
 * stage 0 = FAST candidates of (frame, level) before the quadtree, in the reference's emission order,
 *           coordinates relative to (16,16) as at ORBextractor.cc:796-803;
 * stage 1 = key-points of the level after DistributeOctTree + IC_Angle, level coordinates. */
int rumi_orb_stage_keypoints(RumiOrb *h, int32_t frame, int32_t level, int32_t stage, RumiKeyPoint *out,
                             int32_t cap, int32_t *n_out);

/* Average device time per stage of the last batched call, in milliseconds (HIP events on the call's
 * stream): [0] pyramid, [1] FAST+NMS cells, [2] candidate compaction, [3] blur, [4] quadtree,
 * [5] orientation+descriptors, [6] total.  Only filled when rumi_orb_set_profiling(h, 1) was called. */
int rumi_orb_set_profiling(RumiOrb *h, int32_t on);
int rumi_orb_stage_ms(RumiOrb *h, float ms[8]);

#ifdef __cplusplus
}
#endif
#endif /* RUMI_ORB_H */
