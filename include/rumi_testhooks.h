/*
 * rumi_testhooks.h — host-only entry points of librumi_hip.so that expose, for CPU tests, the pieces of
 * product code that are shared between host and device builds (the same source compiles both ways):
 * the replay of libstdc++'s std::sort used by the quadtree, the array quadtree itself, and the scalar
 * math of orb_math.h.  None of these touch a GPU; none is used by the reference-facing API.
 */
#ifndef RUMI_TESTHOOKS_H
#define RUMI_TESTHOOKS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Sort (key, id) pairs exactly as std::sort(first,last,compareNodes) of libstdc++ would
 * (ORBextractor.cc:524-536,658); ids are permuted in place alongside keys. */
int rumi_hook_sort_like_std(uint32_t *keys, uint16_t *ids, int32_t n);
/* The same through the workgroup-parallel replay on the GPU (n <= 4096; needs a device) and through the real std::sort. */
int rumi_hook_sort_device(uint32_t *keys, uint16_t *ids, int32_t n);
int rumi_hook_std_sort(uint32_t *keys, uint16_t *ids, int32_t n);

/* DistributeOctTree (ORBextractor.cc:538-724) on packed candidates x | y<<12 | score<<24 (coordinates
 * relative to (minX,minY)); writes indices into `cand` in the reference's result order. */
int rumi_hook_quadtree(const uint32_t *cand, int32_t n, int32_t minX, int32_t maxX, int32_t minY, int32_t maxY,
                       int32_t N, int32_t *out_idx, int32_t cap, int32_t *n_out);

float rumi_hook_sinf(float x);           /* restated glibc sinf  (orb_math.h) */
float rumi_hook_cosf(float x);           /* restated glibc cosf  (orb_math.h) */
float rumi_hook_fast_atan2(float y, float x);   /* cv::fastAtan2, degrees */
int rumi_hook_cv_round(float v);         /* cvRound */
int rumi_hook_magic_div(int32_t idx, int32_t d);   /* divide-free idx / d used by the FAST cell kernel (orb_geom.h) */

#ifdef __cplusplus
}
#endif
#endif
