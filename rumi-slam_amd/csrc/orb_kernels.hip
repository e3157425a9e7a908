// HIP kernels of the ORB front-end for gfx950 (wave64).  See DESIGN.md for the data layout and the
// roofline of each kernel.  Built with -ffp-contract=off: the float steering math of rBRIEF and the
// atan polynomial must round exactly like the reference's x86-64 SSE code.
//
// Reference behaviour restated per kernel (R/ = /root/reference/src/rumi-slam/):
//   k_resize        cv::resize INTER_LINEAR 8U, level l from level l-1   R/lib_src/ORBextractor.cc:1103
//   k_fast_cells    per-cell cv::FAST(iniTh | minTh, NMS)                R/lib_src/ORBextractor.cc:748-807
//   k_compact       concatenation of the cell results in cell order      R/lib_src/ORBextractor.cc:796-803
//   k_blur          cv::GaussianBlur 7x7 sigma 2, REFLECT_101             R/lib_src/ORBextractor.cc:1057-1058
//   k_orient_desc   IC_Angle + computeOrbDescriptor + output assembly    R/lib_src/ORBextractor.cc:73-143,1067-1088
#include <hip/hip_runtime.h>

#include "orb_device.h"
#include "orb_math.h"

namespace rumi {

__constant__ int8_t c_pattern[256 * 4] = {
#include "orb_pattern.inc"
};

// XCD-aware workgroup placement (cdna_hip_programming.md T1): the dispatcher deals consecutive workgroups round-robin over
// the 8 XCDs, each with a private L2.  Remapping the linear workgroup id with this bijection gives every XCD one contiguous
// range of logical ids, so neighbouring tiles of one frame (which share 64-B lines and halo rows) meet in the same L2.
// Placement only changes speed / HBM traffic, never results.
__device__ __forceinline__ unsigned xcd_swizzle(unsigned lin, unsigned total) {
    const unsigned q = total >> 3, r = total & 7, xcd = lin & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
}

struct __attribute__((packed)) U64 { uint64_t v; };      // possibly unaligned 8-byte global access

// pixel (0,0) of a pyramid level; a REFLECT_101 frame of kPadX x kPadY pixels surrounds it in memory
__device__ __forceinline__ const uint8_t *level_base(const ImgSrc &s, const DevParams *P, int level, int frame,
                                                      int *pitch) {
    *pitch = P->lv[level].pitch;
    return s.pyr + (long long)frame * P->arenaStride + P->lv[level].off;
}

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

// ------------------------------------------------------------------------------------------------
// Pyramid.  Every level is stored with the reference's BORDER_REFLECT_101 frame (copyMakeBorder,
// ORBextractor.cc:1105-1110) so that the blur and every gather run without edge cases.
//   k_pyr0    level 0 = the caller's frame + frame; a lane moves 4 pixels.
//   k_resize  level l from level l-1 (cv::resize INTER_LINEAR 8U; taps from host tables that follow cv's coefficient
//             rule, orb_geom.h); a lane produces 4 horizontally adjacent pixels of the FRAMED output (frame pixels
//             recompute their mirror pixel) and stores one dword.  HBM-bound: 1.19 B written, ~1.44 B read per pixel.
// ------------------------------------------------------------------------------------------------
// k_pyr0 / k_resize write the pixel columns [0, w) of the rows [-kFrameRows, h + kFrameRows) (the frame rows above and below are
// computed like any other row, from the mirrored source row); k_frame_cols then mirrors the left / right frame columns of
// every level of every frame in ONE launch (no level reads another level's frame columns).
__global__ __launch_bounds__(256) void k_pyr0(const DevParams *__restrict__ P, ImgSrc src) {
    const DevLevel &D = P->lv[0];
    const unsigned wg = xcd_swizzle((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, gridDim.x * gridDim.y * gridDim.z);
    const int bx = wg % gridDim.x, by = (wg / gridDim.x) % gridDim.y, frame = wg / (gridDim.x * gridDim.y);
    const int ox = (bx * 64 + (threadIdx.x & 63)) * 4;
    const int oy = by * 4 + (threadIdx.x >> 6) - kFrameRows;
    if (ox >= D.w || oy >= D.h + kFrameRows) return;
    const uint8_t *in = src.l0 + (long long)frame * src.l0FrameStride + (long long)reflect101(oy, D.h) * src.l0Pitch + ox;
    uint8_t *out = src.pyr + (long long)frame * P->arenaStride + D.off + (long long)oy * D.pitch + ox;
    uint32_t v;
    if (ox + 3 < D.w && (reinterpret_cast<uintptr_t>(in) & 3) == 0) v = *reinterpret_cast<const uint32_t *>(in);
    else {
        v = in[0];
        if (ox + 1 < D.w) v |= (uint32_t)in[1] << 8;
        if (ox + 2 < D.w) v |= (uint32_t)in[2] << 16;
        if (ox + 3 < D.w) v |= (uint32_t)in[3] << 24;
    }
    *reinterpret_cast<uint32_t *>(out) = v;                                  // columns >= w are rewritten by k_frame_cols
}

// Each lane produces 4 pixels of kResizeRows consecutive rows: the column tables are loaded once and the 2 x kResizeRows
// source-row loads are issued back to back, so a wave has 8 x more bytes in flight per dependent round trip than with one row
// (the kernel is latency-bound: two dependent table -> pixel round trips per wave).
constexpr int kResizeRows = 4;
__global__ __launch_bounds__(256) void k_resize(const DevParams *__restrict__ P, ImgSrc src,
                                                const int16_t *__restrict__ coef, const RowTap *__restrict__ rowTab, int level) {
    const DevLevel &D = P->lv[level];
    const DevLevel &S = P->lv[level - 1];
    const unsigned wg = xcd_swizzle((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, gridDim.x * gridDim.y * gridDim.z);
    const int bx = wg % gridDim.x, by = (wg / gridDim.x) % gridDim.y, frame = wg / (gridDim.x * gridDim.y);
    const int ox = (bx * 64 + (threadIdx.x & 63)) * 4;
    const int oyBase = (by * 4 + (threadIdx.x >> 6)) * kResizeRows - kFrameRows;
    if (ox >= D.w || oyBase >= D.h + kFrameRows) return;
    const uint8_t *sb = src.pyr + (long long)frame * P->arenaStride + S.off;
    uint8_t *dbase = src.pyr + (long long)frame * P->arenaStride + D.off + ox;
    const int16_t *xofs = coef + D.coefX, *xa = xofs + D.w;
    // per output row (frame rows included): the two source rows and the vertical taps come ready from a host-built table (the mirrored
    // row, the clamps and the row pitch products are the same for every lane of every frame)
    const uint8_t *r0p[kResizeRows], *r1p[kResizeRows];
    uint32_t bh0[kResizeRows], bh1[kResizeRows];
    bool live[kResizeRows];
#pragma unroll
    for (int r = 0; r < kResizeRows; r++) {
        const int oy = oyBase + r;
        live[r] = oy < D.h + kFrameRows;
        const RowTap t = rowTab[D.rowTab + (live[r] ? oy : 0) + kFrameRows];
        r0p[r] = sb + t.off0; r1p[r] = sb + t.off1;
        bh0[r] = t.bh0; bh1[r] = t.bh1;
    }
    const bool whole = ox + 3 < D.w;
    const int sx0 = xofs[ox];
    if (whole && ox + 3 < D.xmax && xofs[ox + 3] + 1 - sx0 <= 7) {
        // the 4 outputs read source bytes sx0 .. sx0+7 of two rows -> two (unaligned) 8-byte loads per row; offsets and taps
        // come as one 8-byte and one 16-byte table load
        const uint64_t ofs = reinterpret_cast<const U64 *>(xofs + ox)->v;
        const U64 *t8 = reinterpret_cast<const U64 *>(xa + 2 * ox);
        const uint64_t ta = t8[0].v, tb = t8[1].v;
        uint64_t s0[kResizeRows], s1[kResizeRows];
#pragma unroll
        for (int r = 0; r < kResizeRows; r++) {
            s0[r] = reinterpret_cast<const U64 *>(r0p[r] + sx0)->v;
            s1[r] = reinterpret_cast<const U64 *>(r1p[r] + sx0)->v;
        }
        // horizontal pass as a 2-element dot product: the two source bytes of an output are adjacent, v_perm_b32 spreads them into
        // 16-bit halves and v_dot2_u32_u16 multiplies by the (non-negative, <= 2048) tap pair as it lies in the table
        typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));
        // output i reads the source bytes k_i, k_i + 1 of the 8-byte window: ONE v_perm_b32 over the window's two dwords puts them into the
        // 16-bit halves [b0, 0, b1, 0] (selector built once per column, used for 2 source rows x kResizeRows outputs)
        uint32_t sel[4], tap[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t k = (uint32_t)((int)(int16_t)(ofs >> (16 * i)) - sx0);          // 0 .. 6
            sel[i] = k | 0x0c000c00u | ((k + 1u) << 16);
            const uint64_t tt = i < 2 ? ta : tb;
            tap[i] = (uint32_t)(tt >> (32 * (i & 1)));
        }
#pragma unroll
        for (int r = 0; r < kResizeRows; r++) {
            // vertical taps come pre-shifted: (b * x) >> 16 == mulhi(b << 16, x) for the non-negative operands here (b <= 2048, x <= 32 640)
            uint32_t packed = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t p0 = __builtin_amdgcn_perm((uint32_t)(s0[r] >> 32), (uint32_t)s0[r], sel[i]);
                const uint32_t p1 = __builtin_amdgcn_perm((uint32_t)(s1[r] >> 32), (uint32_t)s1[r], sel[i]);
                const uint32_t q0 = __builtin_amdgcn_udot2(__builtin_bit_cast(v2u16, p0), __builtin_bit_cast(v2u16, tap[i]), 0u, false);
                const uint32_t q1 = __builtin_amdgcn_udot2(__builtin_bit_cast(v2u16, p1), __builtin_bit_cast(v2u16, tap[i]), 0u, false);
                packed |= ((__umulhi(bh0[r], q0 >> 4) + __umulhi(bh1[r], q1 >> 4) + 2u) >> 2) << (8 * i);
            }
            if (live[r]) *reinterpret_cast<uint32_t *>(dbase + (long long)(oyBase + r) * D.pitch) = packed;
        }
    } else {
        for (int r = 0; r < kResizeRows; r++) {
            if (!live[r]) continue;
            uint32_t packed = 0;
            for (int i = 0; i < 4 && ox + i < D.w; i++) {
                const int dx = ox + i;
                const int sx = xofs[dx];
                int q0, q1;
                if (dx < D.xmax) {
                    const int a0 = xa[dx * 2], a1 = xa[dx * 2 + 1];
                    q0 = r0p[r][sx] * a0 + r0p[r][sx + 1] * a1;
                    q1 = r1p[r][sx] * a0 + r1p[r][sx + 1] * a1;
                } else {
                    q0 = r0p[r][sx] * 2048;
                    q1 = r1p[r][sx] * 2048;
                }
                packed |= (uint32_t)(((((int)(bh0[r] >> 16) * (q0 >> 4)) >> 16) + (((int)(bh1[r] >> 16) * (q1 >> 4)) >> 16) + 2) >> 2) << (8 * i);
            }
            *reinterpret_cast<uint32_t *>(dbase + (long long)(oyBase + r) * D.pitch) = packed;
        }
    }
}

// left / right REFLECT_101 frame columns of all levels: thread = (frame, level, framed row, one of 3 aligned dword strips)
__global__ __launch_bounds__(256) void k_frame_cols(const DevParams *__restrict__ P, ImgSrc src) {
    const int level = blockIdx.y, frame = blockIdx.z;
    const DevLevel &D = P->lv[level];
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int strip = t % 3, row = t / 3 - kFrameRows;
    if (row >= D.h + kFrameRows) return;
    uint8_t *rp = src.pyr + (long long)frame * P->arenaStride + D.off + (long long)row * D.pitch;
    // strip 0: x = -4..-1; strips 1, 2: x = (w & ~3) .. (w & ~3) + 7 (covers w .. w+3 and re-writes <= 3 interior pixels)
    const int x0 = strip == 0 ? -kFrameCols : (D.w & ~3) + 4 * (strip - 1);
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) v |= (uint32_t)rp[reflect101(x0 + i, D.w)] << (8 * i);
    *reinterpret_cast<uint32_t *>(rp + x0) = v;
}

// ------------------------------------------------------------------------------------------------
// FAST 9/16 + per-cell NMS + threshold fallback, one wave per (frame, cell).
//
// score(p) = max over the 16 arcs of 9 of min |v - I_k| on the bright or dark side, minus 1  (cv's
// cornerScore with the start threshold folded out); p is a corner at T  <=>  score(p) >= T, so ONE
// score tile serves both thresholds.  NMS neighbours outside the cell's detection region count as 0,
// exactly as cv::FAST's zero-initialised score rows make them.
// The sub-image is staged in LDS; scores never touch HBM.  A cheap necessary test on every pixel selects the (pixel, polarity)
// pairs that get the exact score (fast_quick_pair / fast_score_polar).
// ------------------------------------------------------------------------------------------------
#include <algorithm>
__device__ __forceinline__ int min3i(int a, int b, int c) { return min(min(a, b), c); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }

// One WAVE per (frame, cell), four cells per 256-thread workgroup, no workgroup barrier anywhere: the wave stages its
// sub-image as aligned dwords, scores 64 pixels per step, and emits in index order with a running offset.  LDS per wave is
// sized by the host from the largest cell of the current geometry (FastLds), so occupancy is not limited by LDS.
struct FastLds { int tp, sp, tileBytes, scBytes, maxIters, perWave; };

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// FAST 9/16 corner score of the pixel at t (tile pitch TP) for ONE polarity: sign = +1 scores the "darker ring" arcs
// (contrast v - p), sign = -1 the "brighter ring" arcs (p - v).  cv::cornerScore<16> = max of the two, minus 1: max over the 16
// arcs of 9 contiguous circle pixels of the minimum contrast (a pixel is a corner at threshold T iff score >= T).
__device__ __forceinline__ int fast_score_polar(const uint8_t *t, const int TP, const int sign) {
    const int sv = sign * (int)t[0], ns = -sign;
    int d[16];
    d[0] = sv + ns * t[3 * TP];       d[1] = sv + ns * t[3 * TP + 1];   d[2] = sv + ns * t[2 * TP + 2];   d[3] = sv + ns * t[TP + 3];
    d[4] = sv + ns * t[3];            d[5] = sv + ns * t[-TP + 3];      d[6] = sv + ns * t[-2 * TP + 2];  d[7] = sv + ns * t[-3 * TP + 1];
    d[8] = sv + ns * t[-3 * TP];      d[9] = sv + ns * t[-3 * TP - 1];  d[10] = sv + ns * t[-2 * TP - 2]; d[11] = sv + ns * t[-TP - 3];
    d[12] = sv + ns * t[-3];          d[13] = sv + ns * t[TP - 3];      d[14] = sv + ns * t[2 * TP - 2];  d[15] = sv + ns * t[3 * TP - 1];
    int lo3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) lo3[k] = min3i(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
    int A = -256;
#pragma unroll
    for (int k = 0; k < 16; k += 2)
        A = max3i(A, min3i(lo3[k], lo3[(k + 3) & 15], lo3[(k + 6) & 15]), min3i(lo3[k + 1], lo3[(k + 4) & 15], lo3[(k + 7) & 15]));
    return A - 1;
}

// Necessary condition for score >= thr - 1, per polarity: every arc of 9 contains 4 consecutive of the 8 EVEN circle positions,
// so the best "4 consecutive even positions" contrast bounds that polarity's score from above.  Evaluated for TWO horizontally
// adjacent pixels per lane in packed 16-bit halves (v_pk_sub/min/max_i16: contrasts are in [-255, 255]).  Returns bit 0 / bit 1 =
// the darker-ring polarity of pixel 0 / pixel 1 can reach thr, bit 2 / bit 3 = the brighter-ring polarity can.  Pixels with no
// bit set can never reach the threshold; the others are compacted (one entry per polarity) and scored exactly for that polarity.
typedef short v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2s pair_at(const uint8_t *t, const int o) { return __builtin_bit_cast(v2s, (uint32_t)t[o] | ((uint32_t)t[o + 1] << 16)); }
__device__ __forceinline__ int fast_quick_pair(const uint8_t *t, const int TP, const int thr, int &probe) {
    const v2s v = pair_at(t, 0);
    v2s e[8];
    e[0] = v - pair_at(t, 3 * TP);  e[2] = v - pair_at(t, 3);   e[4] = v - pair_at(t, -3 * TP); e[6] = v - pair_at(t, -3);
    // four consecutive even positions always include two ADJACENT compass positions (0, 4, 8, 12): when no lane of the wave has such a
    // pair beyond the threshold on either side, the other four loads and the arc search are skipped for all 128 pixels (flat image regions).
    // The pre-test costs a fifth of the full test, so a cell only keeps running it while it pays (probe)
    if (probe > 0) {                                  // probe: 3 = undecided (counts down on every miss), 4 = the pre-test has skipped at least once in this cell, 0 = given up
        const v2s a = __builtin_elementwise_min(e[0], e[2]), b = __builtin_elementwise_min(e[2], e[4]), c = __builtin_elementwise_min(e[4], e[6]),
                  d = __builtin_elementwise_min(e[6], e[0]);
        const v2s A4 = __builtin_elementwise_max(__builtin_elementwise_max(a, b), __builtin_elementwise_max(c, d));
        const v2s a2 = __builtin_elementwise_max(e[0], e[2]), b2 = __builtin_elementwise_max(e[2], e[4]), c2 = __builtin_elementwise_max(e[4], e[6]),
                  d2 = __builtin_elementwise_max(e[6], e[0]);
        const v2s B4 = __builtin_elementwise_min(__builtin_elementwise_min(a2, b2), __builtin_elementwise_min(c2, d2));
        const v2s th = {(short)thr, (short)thr}, th1 = {(short)(thr - 1), (short)(thr - 1)};
        const uint32_t da = ~__builtin_bit_cast(uint32_t, A4 - th), db = __builtin_bit_cast(uint32_t, B4 + th1);
        if (__ballot(((da | db) & 0x80008000u) != 0) == 0) { probe = 4; return 0; }
        if (probe < 4) probe--;                       // a cell whose first three steps never skip is textured: stop paying for the pre-test
    }
    e[1] = v - pair_at(t, 2 * TP + 2);  e[3] = v - pair_at(t, -2 * TP + 2); e[5] = v - pair_at(t, -2 * TP - 2); e[7] = v - pair_at(t, 2 * TP - 2);
    v2s lo2[8], hi2[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { lo2[k] = __builtin_elementwise_min(e[k], e[(k + 1) & 7]); hi2[k] = __builtin_elementwise_max(e[k], e[(k + 1) & 7]); }
    v2s A = __builtin_elementwise_min(lo2[0], lo2[2]), Bn = __builtin_elementwise_max(hi2[0], hi2[2]);
#pragma unroll
    for (int k = 1; k < 8; k++) {
        A = __builtin_elementwise_max(A, __builtin_elementwise_min(lo2[k], lo2[(k + 2) & 7]));
        Bn = __builtin_elementwise_min(Bn, __builtin_elementwise_max(hi2[k], hi2[(k + 2) & 7]));
    }
    // A >= thr  <=>  sign(A - thr) clear;   -Bn >= thr  <=>  Bn + (thr - 1) < 0  <=>  sign set
    const v2s th = {(short)thr, (short)thr}, th1 = {(short)(thr - 1), (short)(thr - 1)};
    const uint32_t da = ~__builtin_bit_cast(uint32_t, A - th), db = __builtin_bit_cast(uint32_t, Bn + th1);
    return (int)(((da >> 15) & 1u) | ((da >> 30) & 2u) | ((db >> 13) & 4u) | ((db >> 28) & 8u));
}

// exact scores of up to 64 ring entries (entry = pixel index | polarity << 15).  A pixel that passed both quick tests has two
// entries, darker first; darker entries store their score, then brighter entries keep the maximum (their darker twin sits
// earlier in the ring, i.e. in this batch or a previous one).
// Every pixel whose score reaches tlow is also appended to the cell's SCORED LIST sl (pixel indices, ascending because the ring is
// filled in pixel order; a pixel scored for both polarities appears twice in a row): NMS and emission then walk a few hundred
// listed pixels instead of the whole score map.  nScored counts all appends; entries beyond kScoredCap are dropped and the caller
// falls back to scanning the map.
constexpr int kScoredCap = 512;
template <int CTP>
__device__ __forceinline__ void fast_score_batch(const uint8_t *tile, uint8_t *sc, uint16_t *sl, int &nScored, int entry, bool active, int tp, int SP,
                                                 int shx, int dw, unsigned Mdw, int tlow, int lane) {
    const int TP = CTP ? CTP : tp;
    const int i2 = entry & 0x7FFF, bright = entry >> 15;
    const int py = magic_div(i2, Mdw), px = i2 - mul24(py, dw);
    int s = 0;
    if (active) s = fast_score_polar(&tile[mul24(py + 3, TP) + px + 3 + shx], TP, bright ? -1 : 1);
    uint8_t *dst = &sc[mul24(py + 1, SP) + px + 1];
    const bool hit = active && s >= tlow;
    if (hit && !bright) *dst = (uint8_t)s;
    const unsigned long long bh = __ballot(hit);
    const int pos = nScored + __popcll(bh & ((1ull << lane) - 1ull));
    if (hit && pos < kScoredCap) sl[pos] = (uint16_t)i2;
    nScored += __popcll(bh);
    wave_lds_fence();
    if (hit && bright && s > (int)*dst) *dst = (uint8_t)s;
}

// score map of one cell: quick test on every pixel (two per lane), exact score on the compacted survivors (CTP != 0: compile-time
// tile pitch).  Ring entries = pixel index | polarity << 15; a pixel's darker entry always precedes its brighter one.
template <int CTP>
__device__ __forceinline__ int fast_score_cell(const uint8_t *tile, uint8_t *sc, uint16_t *cl, uint16_t *sl, int tp, int SP, int shx, int dw, int dh,
                                               unsigned Mdw, int tlow, int lane) {
    const int TP = CTP ? CTP : tp;
    const int pw = (dw + 1) >> 1, npairs = pw * dh;       // pixel pairs per row / per cell (the last pair of an odd row is half empty)
    const unsigned Mpw = magic_of(pw);
    int head = 0, pending = 0;                         // circular ring: entries wait in cl[(head + k) & 511], k < pending (< 64 between steps)
    int nScored = 0, probe = 3;
    for (int base = 0; base < npairs; base += 64) {
        const int ip = base + lane;
        int pass = 0, idx = 0;
        if (ip < npairs) {
            const int py = magic_div(ip, Mpw), px = (ip - mul24(py, pw)) * 2;
            idx = mul24(py, dw) + px;
            pass = fast_quick_pair(&tile[mul24(py + 3, TP) + px + 3 + shx], TP, tlow + 1, probe);
            if (px + 1 >= dw) pass &= 5;                // second pixel of the pair lies outside the detection region
        }
        if (probe == 4 && __ballot(pass != 0) == 0) continue;   // flat stretch of a cell that has them: nothing to append (wave-uniform)
        // ring positions: entries of lower lanes first; within a lane darker(px0), brighter(px0), darker(px1), brighter(px1).  A lane adds
        // 0..4 entries: the exclusive prefix of that count over the lanes comes from three bit-plane ballots (v_mbcnt) instead of four
        // per-flag ballots with a masked popcount each
        const int cnt = __popc(pass);
        const unsigned long long c0 = __ballot(cnt & 1), c1 = __ballot(cnt & 2), c2 = __ballot(cnt & 4);
        const int p0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(c0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)c0, 0u));
        const int p1 = __builtin_amdgcn_mbcnt_hi((uint32_t)(c1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)c1, 0u));
        const int p2 = __builtin_amdgcn_mbcnt_hi((uint32_t)(c2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)c2, 0u));
        int pos = head + pending + p0 + 2 * p1 + 4 * p2;
        if (pass & 1) cl[pos++ & 511] = (uint16_t)idx;
        if (pass & 4) cl[pos++ & 511] = (uint16_t)(idx | 0x8000);
        if (pass & 2) cl[pos++ & 511] = (uint16_t)(idx + 1);
        if (pass & 8) cl[pos & 511] = (uint16_t)((idx + 1) | 0x8000);
        pending += __popcll(c0) + 2 * __popcll(c1) + 4 * __popcll(c2);
        while (pending >= 64) {                        // a full wave of entries: score them exactly
            wave_lds_fence();
            const int e = cl[(head + lane) & 511];
            fast_score_batch<CTP>(tile, sc, sl, nScored, e, true, tp, SP, shx, dw, Mdw, tlow, lane);
            head = (head + 64) & 511;
            pending -= 64;
        }
    }
    wave_lds_fence();
    fast_score_batch<CTP>(tile, sc, sl, nScored, lane < pending ? cl[(head + lane) & 511] : 0, lane < pending, tp, SP, shx, dw, Mdw, tlow, lane);
    return nScored;
}

// TPC / SPC: tile and score-map pitches as compile-time constants: the sixteen circle offsets and the NMS neighbours then are immediate LDS
// offsets instead of one address add each (-1 % kernel time; rounding the pitches up to 64 instead costs +13 %: LDS footprint); 0 = run-time
template <int TPC, int SPC>
__global__ __launch_bounds__(256) void k_fast_cells(const DevParams *__restrict__ P, ImgSrc src, FastLds F,
                                                    uint32_t *__restrict__ cellBuf, int32_t *__restrict__ cellCnt) {
    extern __shared__ __attribute__((aligned(16))) uint8_t fl[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned wg = xcd_swizzle(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int cell = (wg % gridDim.x) * 4 + wave, frame = wg / gridDim.x;
    if (cell >= P->totalCells) return;
    uint8_t *tile = fl + (size_t)wave * F.perWave;
    uint8_t *sc = tile + F.tileBytes;
    unsigned long long *balI = reinterpret_cast<unsigned long long *>(sc + F.scBytes), *balM = balI + F.maxIters;
    int level = 0;
    for (int l = 1; l < P->nlevels; l++)
        if (cell >= P->lv[l].cellBase) level = l;
    const DevLevel &L = P->lv[level];
    const int ci = cell - L.cellBase;
    const int ci_i = ci / L.nCols, ci_j = ci - ci_i * L.nCols;
    const long long cellIdx = (long long)frame * P->totalCells + cell;
    const int iniY = kBorder + ci_i * L.hCell, iniX = kBorder + ci_j * L.wCell;
    const int maxY = min(iniY + L.hCell + 6, L.maxBY), maxX = min(iniX + L.wCell + 6, L.maxBX);
    const int cols = maxX - iniX, rows = maxY - iniY;
    // skip rules of ORBextractor.cc:752,760 and cv::FAST's 3-px margins
    if (iniY >= L.maxBY - 3 || iniX >= L.maxBX - 6 || cols < 7 || rows < 7) {
        if (lane == 0) cellCnt[cellIdx] = 0;
        return;
    }
    const int TP = TPC ? TPC : F.tp, SP = SPC ? SPC : F.sp;
    int pitch;
    const int shx = iniX & 3;
    const uint8_t *img = level_base(src, P, level, frame, &pitch) + (long long)iniY * pitch + (iniX - shx);
    const int nd = (shx + cols + 3) >> 2;
    const unsigned Mnd = magic_of(nd);
    for (int idx = lane; idx < rows * nd; idx += 64) {
        const int r = magic_div(idx, Mnd), c = idx - mul24(r, nd);
        *reinterpret_cast<uint32_t *>(&tile[mul24(r, TP) + 4 * c]) = *reinterpret_cast<const uint32_t *>(img + mul24(r, pitch) + 4 * c);
    }
    const int dw = cols - 6, dh = rows - 6;
    const unsigned Mdw = magic_of(dw);
    for (int idx = lane * 4; idx < (dh + 2) * SP; idx += 256) *reinterpret_cast<uint32_t *>(&sc[idx]) = 0;
    wave_lds_fence();
    const int npx = dw * dh;
    uint16_t *cl = reinterpret_cast<uint16_t *>(balM + F.maxIters);
    uint16_t *sl = cl + 512;
    // Two passes, as upstream calls cv::FAST (:771-785): threshold iniThFAST first, and minThFAST only when the cell yields no key-point (after
    // NMS) at iniThFAST.  A pixel below the pass's threshold can neither be emitted nor suppress a neighbour (cv::FAST's score rows hold 0
    // for it, and NMS needs a strictly larger neighbour), so each pass scores only what reaches ITS threshold: at iniThFAST the quick test
    // passes a fraction of the pixels it passes at minThFAST, and textured cells never run the second pass.
    int thr = max(1, P->iniTh);
    int nScored, nItems, iters, found;
    bool listed;
#pragma nounroll
    for (int pass = 0;; pass++) {
        switch (TP) {                                                // compile-time pitches for the common geometries
            case 48: nScored = fast_score_cell<48>(tile, sc, cl, sl, TP, SP, shx, dw, dh, Mdw, thr, lane); break;
            case 52: nScored = fast_score_cell<52>(tile, sc, cl, sl, TP, SP, shx, dw, dh, Mdw, thr, lane); break;
            case 56: nScored = fast_score_cell<56>(tile, sc, cl, sl, TP, SP, shx, dw, dh, Mdw, thr, lane); break;
            default: nScored = fast_score_cell<0>(tile, sc, cl, sl, TP, SP, shx, dw, dh, Mdw, thr, lane); break;
        }
        wave_lds_fence();
        // NMS over the scored list (ascending pixel order = the row-major order cv::FAST emits in); a cell with more than
        // kScoredCap scored pixels scans its whole score map instead
        listed = nScored <= kScoredCap;
        nItems = listed ? nScored : npx;
        iters = (nItems + 63) >> 6;
        found = 0;
        for (int it = 0; it < iters; it++) {
            const int k = it * 64 + lane;
            bool isMax = false;
            if (k < nItems) {
                const int idx = listed ? (int)sl[k] : k;
                const bool dup = listed && k > 0 && (int)sl[k - 1] == idx;        // second entry of a pixel scored for both polarities
                const int py = magic_div(idx, Mdw), px = idx - mul24(py, dw);
                const uint8_t *s = &sc[mul24(py + 1, SP) + px + 1];
                const int v = s[0];
                isMax = !dup && v > 0 && v > s[-1] && v > s[1] && v > s[-SP - 1] && v > s[-SP] && v > s[-SP + 1] &&
                        v > s[SP - 1] && v > s[SP] && v > s[SP + 1];
            }
            const unsigned long long bi = __ballot(isMax);
            found += __popcll(bi);
            if (lane == 0) balI[it] = bi;
        }
        wave_lds_fence();
        if (found > 0 || pass == 1) break;                           // retry with minThFAST only if the first call found nothing (:783)
        thr = max(1, P->minTh);                                      // scores of the first pass that are still in the map are rewritten with the same values
    }
    if (lane == 0) cellCnt[cellIdx] = found;
    uint32_t *out = cellBuf + cellIdx * P->maxCellCand;
    int run = 0;
    for (int it = 0; it < iters; it++) {
        const unsigned long long b = balI[it];
        if ((b >> lane) & 1ull) {
            const int k = it * 64 + lane;
            const int idx = listed ? (int)sl[k] : k;
            const int py = magic_div(idx, Mdw), px = idx - mul24(py, dw);
            const uint32_t x = (uint32_t)(px + 3 + ci_j * L.wCell), y = (uint32_t)(py + 3 + ci_i * L.hCell);
            out[run + __popcll(b & ((1ull << lane) - 1ull))] = x | (y << 12) | ((uint32_t)sc[mul24(py + 1, SP) + px + 1] << 24);
        }
        run += __popcll(b);
    }
}

// ------------------------------------------------------------------------------------------------
// Candidate compaction: one workgroup per frame concatenates the cell lists in cell order (levels
// ascending, cells row-major) into cand[frame][...] and writes levelStart[frame][0..nlevels].
// A level that would exceed its capacity is truncated and flagged (levelStart keeps the true count in
// overflow[frame]); the host turns that into RUMI_E_CAPACITY.
// ------------------------------------------------------------------------------------------------
constexpr int kCompactThreads = 1024;
__global__ __launch_bounds__(kCompactThreads) void k_compact(const DevParams *__restrict__ P, const uint32_t *__restrict__ cellBuf,
                                                 const int32_t *__restrict__ cellCnt, uint32_t *__restrict__ cand,
                                                 int32_t *__restrict__ levelStart, int32_t *__restrict__ overflow) {
    extern __shared__ int sStart[];          // totalCells + 1 exclusive prefix
    __shared__ int part[kCompactThreads];
    const int tid = threadIdx.x, frame = blockIdx.x;
    const int nc = P->totalCells;
    const int32_t *cnt = cellCnt + (long long)frame * nc;
    const int chunk = (nc + kCompactThreads - 1) / kCompactThreads;
    int sum = 0;
    for (int k = 0; k < chunk; k++) {
        const int c = tid * chunk + k;
        if (c < nc) sum += cnt[c];
    }
    // exclusive scan of the per-thread sums: shuffles inside a wave, the 16 wave totals through LDS
    const int lane = tid & 63, wave = tid >> 6;
    int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) part[wave] = inc;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kCompactThreads / 64; w++) {
        const int t = part[w];
        if (w < wave) base += t;
        total += t;
    }
    if (tid == 0) sStart[nc] = total;
    int run = base + inc - sum;
    for (int k = 0; k < chunk; k++) {
        const int c = tid * chunk + k;
        if (c < nc) { sStart[c] = run; run += cnt[c]; }
    }
    __syncthreads();
    // few frames per launch: gridDim.y workgroups share a frame's outputs (each repeats the cheap scan), which cuts the latency of a
    // single-frame call; slice 0 publishes the level starts
    int32_t *ls = levelStart + (long long)frame * (kMaxLevels + 1);
    if (blockIdx.y == 0 && tid <= P->nlevels) {
        const int c = tid < P->nlevels ? P->lv[tid].cellBase : nc;
        ls[tid] = sStart[c];
    }
    if (blockIdx.y == 0 && tid < P->nlevels) {
        const int c0 = P->lv[tid].cellBase, c1 = c0 + P->lv[tid].nCells;
        if (sStart[c1] - sStart[c0] > P->lv[tid].candCap) atomicOr(&overflow[frame], 1 << tid);
    }
    uint32_t *out = cand + (long long)frame * P->totalCand;
    // one lane per output element: its cell is the last one whose start is <= j (binary search in the LDS prefix), so every
    // lane has an independent load in flight instead of a wave walking its cells one round trip at a time
    const int nOut = min(sStart[nc], P->totalCand);
    const uint32_t *inBase = cellBuf + (long long)frame * nc * P->maxCellCand;
    for (int j = blockIdx.y * kCompactThreads + tid; j < nOut; j += kCompactThreads * gridDim.y) {
        int lo = 0, hi = nc;                       // invariant: sStart[lo] <= j < sStart[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (sStart[mid] <= j) lo = mid; else hi = mid;
        }
        out[j] = inBase[(long long)lo * P->maxCellCand + (j - sStart[lo])];
    }
}

// ------------------------------------------------------------------------------------------------
// Gaussian blur 7x7, sigma 2, fixed point: taps {18,34,48,56,48,34,18}/256, row pass to u16, column
// pass to u32, (v + 32768) >> 16, REFLECT_101 at the level's own edges.
// ------------------------------------------------------------------------------------------------
// Register formulation: a lane owns a 4-pixel column strip, a wave walks kBlurRows output rows top to bottom.
// Per source row: ONE aligned dword load per lane; the left / right neighbours' dwords arrive by DPP shuffles (the two
// outer lanes load their halo dwords); the 7-tap row pass runs on the 10 unpacked bytes, the column pass on a 7-deep
// register ring of row results; 4 output pixels leave as one dword store.  No LDS, no barriers, and no edge cases: the
// REFLECT_101 frame stored around every level IS the border GaussianBlur(..., BORDER_REFLECT_101) would synthesise.
constexpr int kBlurRows = 16;

// all levels in one launch: workgroup `lin` of a frame belongs to the level whose [base, base + gx * gy) range holds it
struct BlurGrid { int base[kMaxLevels + 1]; int gx[kMaxLevels]; };
// a * b + c on 24-bit operands as ONE v_mad_u32_u24 (the compiler splits the C expression into a multiply and a 3-input add)
__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__global__ __launch_bounds__(256) void k_blur(const DevParams *__restrict__ P, ImgSrc src, BlurGrid G) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned wg = xcd_swizzle(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int frame = wg / gridDim.x, lin = wg % gridDim.x;
    int level = 0;
    for (int l = 1; l < P->nlevels; l++)
        if (lin >= G.base[l]) level = l;
    const DevLevel &L = P->lv[level];
    const int bx = (lin - G.base[level]) % G.gx[level], by = (lin - G.base[level]) / G.gx[level];
    const int xa = bx * 256 + lane * 4;                          // first pixel of my strip
    const int y0 = (by * 4 + wave) * kBlurRows;
    if (y0 >= L.h) return;                                       // whole wave (wave-uniform)
    int pitch;
    const uint8_t *img = level_base(src, P, level, frame, &pitch);
    uint8_t *out = src.blur + (long long)frame * P->arenaStride + L.off;
    const int w = L.w, h = L.h;
    // strips that start beyond the row still feed their neighbours' halos; clamp their address inside the framed row
    const int xl = min(xa, ((w + kPadX - 4) & ~3));
    int ring[7][4];                                              // row results of the last seven source rows; slot = source row mod 7 of this walk
#pragma unroll
    for (int k = 0; k < 7; k++)
#pragma unroll
        for (int i = 0; i < 4; i++) ring[k][i] = 0;
    const int yEnd = min(y0 + kBlurRows, h), rEnd = yEnd + 3;
    // the walk is unrolled by seven so that the ring never moves: source row r0 + j lands in slot j, and the taps of the output row it
    // completes sit at compile-time slots (a runtime ring costs 24 register moves per row)
    const uint8_t *row = img + (long long)(y0 - 3) * pitch - pitch;   // running pointers: one add per row instead of a 64-bit multiply-add
    uint8_t *orow = out + (long long)(y0 - 6) * L.pitch + xa - L.pitch;
    for (int r0 = y0 - 3; r0 < rEnd; r0 += 7) {
#pragma unroll
        for (int j = 0; j < 7; j++) {
            const int r = r0 + j;
            if (r >= rEnd) break;                                // wave-uniform
            row += pitch; orow += L.pitch;                       // rows -3..-1 and h..h+2 are frame rows
            const uint32_t C = *reinterpret_cast<const uint32_t *>(row + xl);
            uint32_t Lw = __shfl_up(C, 1), Rw = __shfl_down(C, 1);
            if (lane == 0) Lw = *reinterpret_cast<const uint32_t *>(row + xl - 4);
            if (lane == 63) Rw = *reinterpret_cast<const uint32_t *>(row + min(xl + 4, (w + kPadX - 4) & ~3));
            // row pass on packed bytes: output i needs the 7 bytes S[i+1 .. i+7] of the 12-byte run {Lw, C, Rw}; two byte-dot-products
            // (v_dot4_u32_u8) against the taps {18,34,48,56} and {48,34,18,0} give the exact integer sum (<= 65 280)
            constexpr uint32_t tA = 18u | (34u << 8) | (48u << 16) | (56u << 24), tB = 48u | (34u << 8) | (18u << 16);
            const uint32_t A0 = __builtin_amdgcn_alignbyte(C, Lw, 1), A1 = __builtin_amdgcn_alignbyte(C, Lw, 2), A2 = __builtin_amdgcn_alignbyte(C, Lw, 3);
            const uint32_t B0 = __builtin_amdgcn_alignbyte(Rw, C, 1), B1 = __builtin_amdgcn_alignbyte(Rw, C, 2), B2 = __builtin_amdgcn_alignbyte(Rw, C, 3);
            ring[j][0] = (int)__builtin_amdgcn_udot4(B0, tB, __builtin_amdgcn_udot4(A0, tA, 0u, false), false);
            ring[j][1] = (int)__builtin_amdgcn_udot4(B1, tB, __builtin_amdgcn_udot4(A1, tA, 0u, false), false);
            ring[j][2] = (int)__builtin_amdgcn_udot4(B2, tB, __builtin_amdgcn_udot4(A2, tA, 0u, false), false);
            ring[j][3] = (int)__builtin_amdgcn_udot4(Rw, tB, __builtin_amdgcn_udot4(C, tA, 0u, false), false);
            const int y = r - 3;                                 // slots (j+1)%7 .. j now hold rows y-3 .. y+3
            if (y >= y0 && xa < w) {
                uint32_t o[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    // rounding constant folded into the first multiply-add; the result's byte 2 is the output pixel (sum <= 255 * 65536 + 32768)
                    // row sums are <= 65 280 and their pairs <= 130 560: 24-bit multiply-adds (v_mad_u32_u24: tap and accumulation in one instruction)
                    uint32_t acc = mad_u24(56u, (uint32_t)ring[(j + 4) % 7][i], 32768u);
                    acc = mad_u24(48u, (uint32_t)(ring[(j + 3) % 7][i] + ring[(j + 5) % 7][i]), acc);
                    acc = mad_u24(34u, (uint32_t)(ring[(j + 2) % 7][i] + ring[(j + 6) % 7][i]), acc);
                    acc = mad_u24(18u, (uint32_t)(ring[(j + 1) % 7][i] + ring[j][i]), acc);
                    o[i] = acc;
                }
                // byte 2 of the four sums -> one dword (v_perm_b32: selectors 0-3 take from the second operand, 4-7 from the first, 0x0c = zero);
                // the blurred arena has the same framed geometry, so a whole dword always fits in the row
                const uint32_t p01 = __builtin_amdgcn_perm(o[1], o[0], 0x0c0c0602u), p23 = __builtin_amdgcn_perm(o[3], o[2], 0x06020c0cu);
                *reinterpret_cast<uint32_t *>(orow) = p01 | p23;         // orow = out + y * pitch + xa
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Orientation + descriptor + output assembly: one HALF wave (32 lanes) per selected key-point, eight key-points per workgroup.
//   The arithmetic that is the same for every lane of a key-point (fastAtan2, the libm sinf / cosf restatement in double precision, the
//   record) is a third of the kernel: with two key-points per wave an instruction serves both.
//   IC_Angle: integer moments over the radius-15 disc of the UN-blurred level, lane = disc column;
//   rBRIEF:   lane l evaluates test pairs l, l+32, ... l+224; __ballot packs 32 bits per key-point at a time, which
//             is exactly the descriptor's little-endian bit order (bit k of byte i = pair 8i+k).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int half_wave_sum(int v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

constexpr int kDiscP = 36, kPatchP = 40;       // LDS row pitches: 31 (+3 alignment slack) and 37 (+3) bytes as whole dwords
constexpr int kKpPerWg = 8;

__global__ __launch_bounds__(256) void k_orient_desc(const DevParams *__restrict__ P, ImgSrc src,
                                                     const uint32_t *__restrict__ selPacked,
                                                     const uint32_t *__restrict__ selMeta,
                                                     const int32_t *__restrict__ selCount, int selCap,
                                                     RumiKeyPoint *__restrict__ kpOut, uint8_t *__restrict__ descOut,
                                                     int outCap) {
    // per key-point: the 31-row disc neighbourhood of the un-blurred level and the 37-row patch of the blurred level, staged
    // with aligned dword loads that are all in flight together (one memory latency instead of 24 dependent byte gathers)
    __shared__ __attribute__((aligned(16))) uint8_t sDisc[kKpPerWg][31 * kDiscP];
    __shared__ __attribute__((aligned(16))) uint8_t sPatch[kKpPerWg][37 * kPatchP];
    const int lane = threadIdx.x & 31, hw = threadIdx.x >> 5;             // lane within the half wave, half-wave index 0..7
    const unsigned wg = xcd_swizzle(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);   // a frame's key-points share one L2
    const int k = (wg % gridDim.x) * kKpPerWg + hw, frame = wg / gridDim.x;
    const bool live = k < selCount[frame];
    int level = 0, slot = 0, x = kEdge, y = kEdge, score = 0;
    if (live) {
        const uint32_t pk = selPacked[(long long)frame * selCap + k], meta = selMeta[(long long)frame * selCap + k];
        level = meta & 0xFF; slot = (int)(meta >> 8);
        x = (int)(pk & 0xFFF) + kBorder; y = (int)((pk >> 12) & 0xFFF) + kBorder; score = (int)(pk >> 24);
    }
    const DevLevel &L = P->lv[level];
    const int xd = (x - kHalfPatch) & ~3, xp = (x - 18) & ~3;     // aligned first columns of the two staged windows
    if (live) {
        int pitch;
        const uint8_t *c = level_base(src, P, level, frame, &pitch) + (long long)(y - kHalfPatch) * pitch + xd;
        const uint8_t *b = src.blur + (long long)frame * P->arenaStride + L.off + (long long)(y - 18) * L.pitch + xp;
        uint32_t vd[9], vp[12];
#pragma unroll
        for (int q = 0; q < 9; q++) {
            const int idx = lane + 32 * q, r = idx / 9, cc = idx - r * 9;
            vd[q] = idx < 31 * 9 ? *reinterpret_cast<const uint32_t *>(c + (long long)r * pitch + 4 * cc) : 0;
        }
#pragma unroll
        for (int q = 0; q < 12; q++) {
            const int idx = lane + 32 * q, r = idx / 10, cc = idx - r * 10;
            vp[q] = idx < 37 * 10 ? *reinterpret_cast<const uint32_t *>(b + (long long)r * L.pitch + 4 * cc) : 0;
        }
#pragma unroll
        for (int q = 0; q < 9; q++) {
            const int idx = lane + 32 * q, r = idx / 9, cc = idx - r * 9;
            if (idx < 31 * 9) *reinterpret_cast<uint32_t *>(&sDisc[hw][r * kDiscP + 4 * cc]) = vd[q];
        }
#pragma unroll
        for (int q = 0; q < 12; q++) {
            const int idx = lane + 32 * q, r = idx / 10, cc = idx - r * 10;
            if (idx < 37 * 10) *reinterpret_cast<uint32_t *>(&sPatch[hw][r * kPatchP + 4 * cc]) = vp[q];
        }
    }
    __syncthreads();
    if (!live) return;

    // IC_Angle (ORBextractor.cc:73-97): lane = column u of the disc; the disc is symmetric (|u| <= umax[|v|]  <=>  |v| <= umax[|u|]), so a
    // lane's rows are |v| <= umax[|u|], known before the loop; m10 = u * (sum of the column), m01 = sum of v * pixel
    const uint8_t *dc = &sDisc[hw][kHalfPatch * kDiscP + (x - xd)];
    const int u = lane - kHalfPatch;
    const int vmaxU = lane < 31 ? P->umax[u < 0 ? -u : u] : -1;
    int colSum = 0, m01 = 0;
#pragma unroll
    for (int i = 0; i < 31; i++) {
        const int v = -kHalfPatch + i;
        if (v <= vmaxU && -v <= vmaxU) {
            const int val = dc[v * kDiscP + u];
            colSum += val;
            m01 += v * val;
        }
    }
    int m10 = u * colSum;
    m10 = half_wave_sum(m10);
    m01 = half_wave_sum(m01);
    const float angle = fast_atan2_deg((float)m01, (float)m10);

    // computeOrbDescriptor (ORBextractor.cc:99-143) on the blurred level
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    const float ang = angle * factorPI;
    const float a = cosf_glibc(ang), b = sinf_glibc(ang);
    const uint8_t *bc = &sPatch[hw][18 * kPatchP + (x - xp)];
    uint32_t bits[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int8_t *pt = &c_pattern[(j * 32 + lane) * 4];
        const float x0 = (float)pt[0], y0 = (float)pt[1], x1 = (float)pt[2], y1 = (float)pt[3];
        const int r0 = cv_round_f(x0 * b + y0 * a), c0 = cv_round_f(x0 * a - y0 * b);
        const int r1 = cv_round_f(x1 * b + y1 * a), c1 = cv_round_f(x1 * a - y1 * b);
        const int t0 = bc[r0 * kPatchP + c0], t1 = bc[r1 * kPatchP + c1];
        const unsigned long long bal = __ballot(t0 < t1);                 // both key-points of the wave; mine is my half
        bits[j] = (uint32_t)(bal >> (32 * (hw & 1)));
    }
    if (slot < outCap) {
        if (lane < 8) {
            uint32_t w = bits[0];
#pragma unroll
            for (int j = 1; j < 8; j++) if (lane == j) w = bits[j];
            reinterpret_cast<uint32_t *>(descOut + ((long long)frame * outCap + slot) * 32)[lane] = w;
        }
        if (lane == 0) {
            RumiKeyPoint kp;
            kp.x = (float)x; kp.y = (float)y;
            if (level != 0) { kp.x = kp.x * L.scale; kp.y = kp.y * L.scale; }   // keypoint->pt *= scale (:1073-1075)
            kp.size = L.patchSize;
            kp.angle = angle;
            kp.response = (float)score;
            kp.octave = level;
            kp.class_id = -1;
            kpOut[(long long)frame * outCap + slot] = kp;
        }
    }
}

// ---- launch wrappers (called from orb_host.hip) ----
void launch_pyr0(const DevParams *dP, const DevParams &hP, ImgSrc src, int nframes, hipStream_t st) {
    dim3 g((hP.lv[0].w + 255) / 256, (hP.lv[0].h + 2 * kFrameRows + 3) / 4, nframes);
    hipLaunchKernelGGL(k_pyr0, g, dim3(256), 0, st, dP, src);
}
void launch_frame_cols(const DevParams *dP, const DevParams &hP, ImgSrc src, int nframes, hipStream_t st) {
    dim3 g(((hP.lv[0].h + 2 * kFrameRows) * 3 + 255) / 256, hP.nlevels, nframes);
    hipLaunchKernelGGL(k_frame_cols, g, dim3(256), 0, st, dP, src);
}
void launch_resize(const DevParams *dP, const DevParams &hP, ImgSrc src, const int16_t *coef, const RowTap *rowTab, int level, int nframes,
                   hipStream_t st) {
    dim3 g((hP.lv[level].w + 255) / 256, (hP.lv[level].h + 2 * kFrameRows + 4 * kResizeRows - 1) / (4 * kResizeRows), nframes);
    hipLaunchKernelGGL(k_resize, g, dim3(256), 0, st, dP, src, coef, rowTab, level);
}
void launch_fast(const DevParams *dP, const DevParams &hP, ImgSrc src, uint32_t *cellBuf, int32_t *cellCnt, int nframes,
                 hipStream_t st) {
    // LDS per wave from the largest cell of this geometry
    int wMax = 0, hMax = 0;
    for (int l = 0; l < hP.nlevels; l++) { wMax = std::max(wMax, hP.lv[l].wCell); hMax = std::max(hMax, hP.lv[l].hCell); }
    FastLds F;
    F.tp = (wMax + 6 + 3 + 3 + 3) & ~3;                   // + alignment shift (<= 3) + dword tail, rounded to 4
    F.sp = (wMax + 2 + 3) & ~3;
    F.tileBytes = (hMax + 6) * F.tp;
    F.scBytes = ((hMax + 2) * F.sp + 15) & ~15;
    F.maxIters = (wMax * hMax + 63) / 64 + 1;
    F.tileBytes = (F.tileBytes + 15) & ~15;
    // tile | score map | two ballot arrays | ring of (pixel, polarity) entries that passed the quick test (circular, 512 x uint16: 63 waiting + up to 4 per lane and step) |
    // list of scored pixels (kScoredCap x uint16)
    F.perWave = (F.tileBytes + F.scBytes + 2 * F.maxIters * 8 + 512 * 2 + 512 * 2 + 15) & ~15;
    // pitches of the common image sizes as compile-time constants (640x480 / 752x480 / 1241x376 / 1024x768: 52, 44; 1280x720: 52, 40;
    // 1920x1080: 48, 40; 848x480: 56, 44; 600x350: 60, 48); anything else takes the run-time instantiation
    const dim3 grid((hP.totalCells + 3) / 4, nframes);
    const size_t lds = (size_t)4 * F.perWave;
#define RUMI_FAST_CASE(T, S) if (F.tp == T && F.sp == S) { hipLaunchKernelGGL((k_fast_cells<T, S>), grid, dim3(256), lds, st, dP, src, F, cellBuf, cellCnt); return; }
    RUMI_FAST_CASE(52, 44) RUMI_FAST_CASE(52, 40) RUMI_FAST_CASE(48, 40) RUMI_FAST_CASE(56, 44) RUMI_FAST_CASE(60, 48)
#undef RUMI_FAST_CASE
    hipLaunchKernelGGL((k_fast_cells<0, 0>), grid, dim3(256), lds, st, dP, src, F, cellBuf, cellCnt);
}
void launch_compact(const DevParams *dP, const DevParams &hP, const uint32_t *cellBuf, const int32_t *cellCnt,
                    uint32_t *cand, int32_t *levelStart, int32_t *overflow, int nframes, hipStream_t st) {
    hipLaunchKernelGGL(k_compact, dim3(nframes, nframes < 32 ? 8 : 1), dim3(kCompactThreads), (hP.totalCells + 1) * sizeof(int), st, dP, cellBuf, cellCnt,
                       cand, levelStart, overflow);
}
void launch_blur(const DevParams *dP, const DevParams &hP, ImgSrc src, int nframes, hipStream_t st) {
    BlurGrid G{};
    int run = 0;
    for (int l = 0; l < hP.nlevels; l++) {
        G.gx[l] = (hP.lv[l].w + 255) / 256;
        G.base[l] = run;
        run += G.gx[l] * ((hP.lv[l].h + 4 * kBlurRows - 1) / (4 * kBlurRows));
    }
    G.base[hP.nlevels] = run;
    hipLaunchKernelGGL(k_blur, dim3(run, nframes), dim3(256), 0, st, dP, src, G);
}
void launch_orient_desc(const DevParams *dP, ImgSrc src, const uint32_t *selPacked, const uint32_t *selMeta,
                        const int32_t *selCount, int selCap, int maxSel, RumiKeyPoint *kpOut, uint8_t *descOut,
                        int outCap, int nframes, hipStream_t st) {
    if (maxSel <= 0) return;
    hipLaunchKernelGGL(k_orient_desc, dim3((maxSel + kKpPerWg - 1) / kKpPerWg, nframes), dim3(256), 0, st, dP, src, selPacked, selMeta,
                       selCount, selCap, kpOut, descOut, outCap);
}

}  // namespace rumi
