/*
 * rumi_queue.h — C ABI of the rumination queue on the GPUs of one node, driven from ONE process.
 *
 * What it replaces in the reference (R/ = /root/reference/src/rumi-slam/): the frames tracking could not use are collected and time-sorted by
 *   CloudImageSampler::TrackStep               R/lib_src/CloudImageSampler.cc:44, the sort at :148-170   (host cv::Mats in mvCurrentCloudProcessImages)
 * and handed on by System::GetCloudProcessImages (R/lib_src/System.cc:1278); KFDSample runs ORBextractor::operator() on them one by one (R/lib_src/KFDSample.cc:113).
 * Here the queue of F frames is cut into contiguous blocks, one per device (block g = frames [g F / n, (g + 1) F / n): the result is already
 * time-ordered), every device extracts its block (rumi_orb_extract_batch_host_records: transfers overlapped with kernels), and ONE exchange step
 * follows: an all-gather of the fixed-capacity per-frame records over RCCL (ncclAllGather inside ncclGroupStart / ncclGroupEnd, one communicator
 * per device from ncclCommInitAll) — after it every device holds the key-points and descriptors of the whole queue and can match any pair of
 * frames locally.  No other collective is on the path.  The reference is one C++ process (System.cc:193-240): so is this.
 *
 * Record of one frame (rumi_queue_record_bytes): { int32 n; int32 monoIndex; RumiKeyPoint kp[cap]; uint8 desc[cap][32] } = 8 + 60 cap bytes.
 * Gathered layout on every device: n_devices blocks of rumi_queue_block_capacity(q) records; block g holds its frames first, then empty records
 * (n = 0) up to the capacity — frame f of a queue of F frames is at row rumi_queue_row(q, F, f).
 *
 * Status codes, error string: rumi_orb.h.  A handle is used by one thread at a time.
 */
#ifndef RUMI_QUEUE_H
#define RUMI_QUEUE_H
#include <stdint.h>

#include "rumi_orb.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct RumiQueue RumiQueue;

/* cfg: the extractor of every shard (cfg->device is ignored; cfg->max_batch = the largest block a shard will see, i.e. ceil(max queue length /
 * n_devices)); devices[n_devices]: HIP device ordinals, one shard each.  An ordinal may REPEAT (several logical shards on one device: the exchange
 * is then made of device-to-device copies — what a one-GPU box can test); with distinct ordinals and n_devices > 1 the exchange is RCCL's
 * (librccl is loaded on first use; RUMI_E_NO_DEVICE if it cannot be).  cap: key-points per record (>= the extractor's worst case, nfeatures + 96
 * is what the other entry points use). */
int rumi_queue_create(const RumiOrbConfig *cfg, const int32_t *devices, int32_t n_devices, int32_t cap, RumiQueue **out);
void rumi_queue_destroy(RumiQueue *q);
int32_t rumi_queue_shards(const RumiQueue *q);
int64_t rumi_queue_record_bytes(const RumiQueue *q);
int32_t rumi_queue_block_capacity(const RumiQueue *q);            /* records per block of the gathered layout (= cfg->max_batch) */
int32_t rumi_queue_row(const RumiQueue *q, int32_t n_frames, int32_t frame);   /* row of `frame` in the gathered layout of a queue of n_frames */
/* 1 if the exchange of this queue is an RCCL all-gather, 0 if device-to-device copies stand in for it (repeated ordinals, or one shard) */
int32_t rumi_queue_uses_rccl(const RumiQueue *q);

/* The whole step: imgs[n_frames] host frames (8-bit grey, `stride` bytes per row, time order), lap0 / lap1 as ORBextractor::operator()'s
 * vLappingArea.  On return every shard's device holds the gathered records; d_gathered[g] (may be NULL as a whole) receives shard g's device pointer
 * (owned by the queue, valid until the next call); h_records (may be NULL): the n_frames records in queue order WITHOUT padding; every shard copies ITS block back
 * from its own device during the extraction (N links instead of one; into PINNED memory sub-chunk by sub-chunk under the kernels, into pageable memory in one
 * copy per shard).  Blocks until everything has arrived.  One host thread per shard feeds the transfers during the extraction; the exchange itself is
 * enqueued by the calling thread. */
int rumi_queue_extract(RumiQueue *q, const uint8_t *const *imgs, int32_t n_frames, int32_t w, int32_t hgt, int32_t stride, int32_t lap0, int32_t lap1,
                       void **d_gathered, uint8_t *h_records);
/* Times of the last call in ms: [0] extraction (slowest shard, wall), [1] exchange (wall, enqueue to completion), [2] 0 (the copy to h_records is part of [0] since round 4), [3] total */
int rumi_queue_last_ms(const RumiQueue *q, float ms[4]);

#ifdef __cplusplus
}
#endif
#endif /* RUMI_QUEUE_H */
