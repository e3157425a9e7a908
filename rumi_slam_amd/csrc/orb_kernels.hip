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

// the 256 rBRIEF test pairs (x0, y0, x1, y1) as floats: a lane fetches its pair with one 16-byte load and no conversions
struct PatternF { float v[256 * 4]; };
constexpr PatternF make_pattern_f() {
    constexpr int8_t src[256 * 4] = {
#include "orb_pattern.inc"
    };
    PatternF p{};
    for (int i = 0; i < 256 * 4; i++) p.v[i] = (float)src[i];
    return p;
}
__constant__ PatternF c_patternF = make_pattern_f();

// XCD-aware workgroup placement (cdna_hip_programming.md T1): the dispatcher deals consecutive workgroups round-robin over
// the 8 XCDs, each with a private L2.  Remapping the linear workgroup id with this bijection gives every XCD one contiguous
// range of logical ids, so neighbouring tiles of one frame (which share 64-B lines and halo rows) meet in the same L2.
// Placement only changes speed / HBM traffic, never results.
__device__ __forceinline__ unsigned xcd_swizzle(unsigned lin, unsigned total) {
    const unsigned q = total >> 3, r = total & 7, xcd = lin & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
}

struct __attribute__((packed)) U32 { uint32_t v; };      // possibly unaligned 4-byte global access
struct __attribute__((packed)) U64 { uint64_t v; };      // possibly unaligned 8-byte global access
constexpr int kPyrLevels = 8, kPyrWinPasses = 20;        // k_pyramid_tiles: levels it is offered for, 4-row passes of its level-0 window

// pixel (0,0) of a pyramid level: level 0 is the caller's frame itself, the others live in the pyramid arena
__device__ __forceinline__ const uint8_t *level_base(const ImgSrc &s, const DevParams *P, int level, int frame,
                                                      int *pitch) {
    if (level == 0) {
        *pitch = s.l0Pitch;
        return s.l0 + (long long)frame * s.l0FrameStride;
    }
    *pitch = P->lv[level].pitch;
    return s.pyr + (long long)frame * P->arenaStride + P->lv[level].off;
}

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

// ------------------------------------------------------------------------------------------------
// Pyramid: level l from level l-1 (cv::resize INTER_LINEAR 8U; taps from host tables that follow cv's coefficient rule, orb_geom.h),
// level 1 straight from the caller's frame.  A lane produces 4 horizontally adjacent pixels of kResizeRows consecutive rows and stores
// one dword per row: the column tables are loaded once and the 2 x kResizeRows source-row loads are issued back to back.
// No border pixels are written: the blur mirrors at the edges itself.
// ------------------------------------------------------------------------------------------------
// (kResizeRows: 4 for the small levels, 8 for levels of 200 rows and more -- launch_resize)
template <int kResizeRows>
__global__ __launch_bounds__(256) void k_resize(const DevParams *__restrict__ P, ImgSrc src,
                                                const int16_t *__restrict__ coef, const RowTap *__restrict__ rowTab, int level, int32_t *__restrict__ clearWord) {
    if (clearWord && (blockIdx.x | blockIdx.y | blockIdx.z | threadIdx.x) == 0) *clearWord = 0;    // the call's error word (orb_host.hip)
    const DevLevel &D = P->lv[level];
    const DevLevel &S = P->lv[level - 1];
    const unsigned wg = xcd_swizzle((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, gridDim.x * gridDim.y * gridDim.z);
    const int bx = wg % gridDim.x, by = (wg / gridDim.x) % gridDim.y, frame = wg / (gridDim.x * gridDim.y);
    const int ox = (bx * 64 + (threadIdx.x & 63)) * 4;
    // (the wave index as a scalar: the row table entries, the source-row pointers and the vertical taps then live in scalar registers)
    const int oyBase = (by * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * kResizeRows;
    if (ox >= D.w || oyBase >= D.h) return;
    int sp;
    const uint8_t *sb = level_base(src, P, level - 1, frame, &sp);
    uint8_t *dbase = src.pyr + (long long)frame * P->arenaStride + D.off + ox;
    const int16_t *xofs = coef + D.coefX, *xa = coef + D.coefXT;
    // per output row: the two source rows and the vertical taps come ready from a host-built table (the clamps are the same for every
    // lane of every frame)
    const uint8_t *r0p[kResizeRows], *r1p[kResizeRows];
    uint32_t bh0[kResizeRows], bh1[kResizeRows];
    bool live[kResizeRows], shared[kResizeRows];                  // shared: the row's first source row is the previous output row's second (scalar)
#pragma unroll
    for (int r = 0; r < kResizeRows; r++) {
        const int oy = oyBase + r;
        live[r] = oy < D.h;
        const RowTap t = rowTab[D.rowTab + (live[r] ? oy : 0)];
        r0p[r] = sb + (long long)t.r0 * sp; r1p[r] = sb + (long long)t.r1 * sp;
        bh0[r] = t.bh0; bh1[r] = t.bh1;
        shared[r] = kResizeRows == 4 && r > 0 && r0p[r] == r1p[r - 1];       // (eight rows a lane: the branches cost 47 registers and the gain, measured)
    }
    const int sx0 = xofs[ox];
    // (the row's last dword may be partial: its surplus outputs come from the padded table entries and land in the row's padding)
    if (ox + 3 < D.xmaxFast && xofs[ox + 3] + 1 - sx0 <= 7) {
        // the 4 outputs read source bytes sx0 .. sx0+7 of two rows -> two (unaligned) 8-byte loads per row; offsets and taps
        // come as one 8-byte and one 16-byte table load.  The window never leaves the source row (the last lanes slide it left).
        const int wx0 = min(sx0, S.w - 8);
        const uint64_t ofs = reinterpret_cast<const U64 *>(xofs + ox)->v;
        const U64 *t8 = reinterpret_cast<const U64 *>(xa + 2 * ox);
        const uint64_t ta = t8[0].v, tb = t8[1].v;
        // (at the usual scale factors five output rows in six start on the source row the row above ended on: that row is neither loaded
        // nor filtered horizontally again -- the test is scalar, the branch is a real one)
        uint64_t s0[kResizeRows], s1[kResizeRows];
#pragma unroll
        for (int r = 0; r < kResizeRows; r++) {
            s0[r] = 0;
            if (!shared[r]) s0[r] = reinterpret_cast<const U64 *>(r0p[r] + wx0)->v;
            s1[r] = reinterpret_cast<const U64 *>(r1p[r] + wx0)->v;
        }
        // horizontal pass as a 2-element dot product: the two source bytes of an output are adjacent, v_perm_b32 spreads them into
        // 16-bit halves and v_dot2_u32_u16 multiplies by the (non-negative, <= 2048) tap pair as it lies in the table
        typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));
        // output i reads the source bytes k_i, k_i + 1 of the 8-byte window: ONE v_perm_b32 over the window's two dwords puts them into the
        // 16-bit halves [b0, 0, b1, 0] (selector built once per column, used for 2 source rows x kResizeRows outputs)
        uint32_t sel[4], tap[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t k = (uint32_t)((int)(int16_t)(ofs >> (16 * i)) - wx0);          // 0 .. 6
            sel[i] = k | 0x0c000c00u | ((k + 1u) << 16);
            const uint64_t tt = i < 2 ? ta : tb;
            tap[i] = (uint32_t)(tt >> (32 * (i & 1)));
        }
        uint32_t hPrev[4] = {0, 0, 0, 0};                              // horizontal results (>> 4) of the previous output row's second source row
#pragma unroll
        for (int r = 0; r < kResizeRows; r++) {
            // vertical taps come pre-shifted: (b * x) >> 16 == mulhi(b << 16, x) for the non-negative operands here (b <= 2048, x <= 32 640)
            uint32_t h0[4], h1[4];
            if (shared[r]) {
#pragma unroll
                for (int i = 0; i < 4; i++) h0[i] = hPrev[i];
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t p0 = __builtin_amdgcn_perm((uint32_t)(s0[r] >> 32), (uint32_t)s0[r], sel[i]);
                    h0[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(v2u16, p0), __builtin_bit_cast(v2u16, tap[i]), 0u, false) >> 4;
                }
            }
            uint32_t packed = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t p1 = __builtin_amdgcn_perm((uint32_t)(s1[r] >> 32), (uint32_t)s1[r], sel[i]);
                h1[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(v2u16, p1), __builtin_bit_cast(v2u16, tap[i]), 0u, false) >> 4;
                packed |= ((__umulhi(bh0[r], h0[i]) + __umulhi(bh1[r], h1[i]) + 2u) >> 2) << (8 * i);
                hPrev[i] = h1[i];
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

// ------------------------------------------------------------------------------------------------
// The pyramid of a call of a few frames in ONE launch.  Seven dependent launches of ~4.4 us are most of such a call's pyramid time, and each
// level is read back from HBM by the next.  Here a workgroup owns a tile of the top level and computes, level by level in LDS, the region of
// every level that tile descends from (plus its share of a partition of the level, so that every pixel of every level is produced): level l - 1's
// region is the source of level l's, the regions (PyrTile, from the host's resize tables) overlap by the taps' reach, and every workgroup
// stores all it computed -- overlapping stores carry the same bytes.  The arithmetic per pixel is k_resize's general path, tap for tap.
// ------------------------------------------------------------------------------------------------
// copyL0: the caller's frame is PINNED HOST memory read over PCIe (a one-frame call without its host-to-device copy): the windows, which cover the frame,
// are stored into the arena's level-0 slot as they pass through the registers, and every later kernel of the call reads level 0 there.
__global__ __launch_bounds__(256) void k_pyramid_tiles(const DevParams *__restrict__ P, ImgSrc src, const int16_t *__restrict__ coef,
                                                       const RowTap *__restrict__ rowTab, const PyrTile *__restrict__ tiles, int bufBytes,
                                                       int32_t *__restrict__ clearWord, int copyL0) {
    // LDS: two image buffers of bufBytes (a level's region and the one computed from it), then the tile's slices of the resize tables
    // (per level and row: source rows relative to the buffer | vertical taps; per level and column: source column relative to the buffer, tap pair)
    extern __shared__ __attribute__((aligned(16))) uint8_t pyrLds[];
    if (clearWord && (blockIdx.x | blockIdx.y | threadIdx.x) == 0) *clearWord = 0;    // the call's error word (orb_host.hip)
    const int frame = blockIdx.y, tid = threadIdx.x, nlevels = P->nlevels;
    uint8_t *A = pyrLds, *B = pyrLds + bufBytes;
    uint2 *tab = reinterpret_cast<uint2 *>(pyrLds + 2 * bufBytes);
    // ---- the per-level parameters first, one lane per level, into LDS: read where they are needed they are a chain of scalar loads from
    // global memory, two or three per level, ~1 us each
    struct Lv { int x0, x1, y0, y1, coefX, coefXT, xmax, rowTab, pitch, pad; long long off; };
    __shared__ Lv sLv[kMaxLevels];
    if (tid < nlevels) {
        const PyrTile &T = tiles[blockIdx.x];
        const DevLevel &D = P->lv[tid];
        sLv[tid] = Lv{T.x0[tid], T.x1[tid], T.y0[tid], T.y1[tid], D.coefX, D.coefXT, D.xmax, D.rowTab, D.pitch, 0, D.off};
    }
    __syncthreads();
    // ---- everything else this workgroup reads from global memory: the table slices of all levels (one row entry and one column entry per
    // thread and level) and the window of level 0 (dwords: 64 columns x 4 rows per pass).  ALL loads are issued before the first value is
    // stored to LDS: as loops of load-then-store they were some fifty dependent round trips, 28 of the kernel's 34 us.
    // (the host offers this kernel for up to kPyrLevels levels, regions of up to 256 rows / columns and windows of up to 80 rows x 256 columns)
    int apitch;
    {
        RowTap rt[kPyrLevels];
        int cofs[kPyrLevels];
        uint32_t ctap[kPyrLevels];
#pragma unroll
        for (int level = 1; level < kPyrLevels; level++) {
            rt[level] = RowTap{0, 0, 0u, 0u}; cofs[level] = 0; ctap[level] = 0;
            if (level < nlevels) {
                const Lv D = sLv[level];
                if (tid < D.y1 - D.y0) rt[level] = rowTab[D.rowTab + D.y0 + tid];
                if (tid < D.x1 - D.x0) {
                    const int dx = D.x0 + tid;
                    cofs[level] = (coef + D.coefX)[dx];
                    // the tap pair as one dword (a0 in the low half); a single-tap column multiplies its one source byte by 2048
                    ctap[level] = dx < D.xmax ? reinterpret_cast<const U32 *>(coef + D.coefXT + dx * 2)->v : 2048u;
                }
            }
        }
        int sp;
        const uint8_t *sb = level_base(src, P, 0, frame, &sp);
        const int ax0 = sLv[0].x0, ay0 = sLv[0].y0, aw = sLv[0].x1 - ax0, ah = sLv[0].y1 - ay0, W0 = P->lv[0].w;
        apitch = (aw + 3) & ~3;
        uint32_t win[kPyrWinPasses];
        const int wx = (tid & 63) * 4, wy = tid >> 6;
#pragma unroll
        for (int k = 0; k < kPyrWinPasses; k++) {
            const int y = wy + 4 * k;
            win[k] = 0;
            if (y < ah && wx < aw) {
                const uint8_t *p = sb + (long long)(ay0 + y) * sp + ax0 + wx;
                if (ax0 + wx + 4 <= W0) win[k] = reinterpret_cast<const U32 *>(p)->v;
                else for (int i = 0; ax0 + wx + i < W0; i++) win[k] |= (uint32_t)p[i] << (8 * i);      // the frame's last columns: no read past the row
            }
        }
        // ---- now the stores
        int tb = 0;
#pragma unroll
        for (int level = 1; level < kPyrLevels; level++) {
            if (level < nlevels) {
                const Lv D = sLv[level];
                const int cols = D.x1 - D.x0, rows = D.y1 - D.y0, sx0 = sLv[level - 1].x0, sy0 = sLv[level - 1].y0;
                if (tid < rows) tab[tb + tid] = make_uint2((uint32_t)(rt[level].r0 - sy0) | ((uint32_t)(rt[level].r1 - sy0) << 16), (rt[level].bh0 >> 16) | (rt[level].bh1 & 0xFFFF0000u));
                if (tid < cols) tab[tb + rows + tid] = make_uint2((uint32_t)(cofs[level] - sx0), ctap[level]);
                tb += rows + cols;
            }
        }
#pragma unroll
        for (int k = 0; k < kPyrWinPasses; k++) {
            const int y = wy + 4 * k;
            if (y < ah && wx < aw) *reinterpret_cast<uint32_t *>(A + y * apitch + wx) = win[k];
        }
        if (copyL0) {
            const DevLevel &D0 = P->lv[0];
            uint8_t *l0 = src.pyr + (long long)frame * P->arenaStride + D0.off;
#pragma unroll
            for (int k = 0; k < kPyrWinPasses; k++) {
                const int y = wy + 4 * k;
                if (y < ah && wx < aw && ax0 + wx + 4 <= D0.pitch) *reinterpret_cast<uint32_t *>(l0 + (long long)(ay0 + y) * D0.pitch + ax0 + wx) = win[k];
            }
        }
    }
    __syncthreads();
    int base = 0;
    for (int level = 1; level < nlevels; level++) {
        const Lv D = sLv[level];
        const int X0 = D.x0, Y0 = D.y0, bw = D.x1 - X0, rows = D.y1 - Y0, gpr = bw >> 2;
        const uint2 *rowT = tab + base, *colT = rowT + rows;
        base += rows + bw;
        uint8_t *dbase = src.pyr + (long long)frame * P->arenaStride + D.off + (long long)Y0 * D.pitch + X0;
        // threads per row of 4-pixel groups: the power of two that holds them (the top levels have eight groups a row)
        const int tprLog = gpr <= 8 ? 3 : gpr <= 16 ? 4 : gpr <= 32 ? 5 : 6, tpr = 1 << tprLog, rstep = 256 >> tprLog;
        for (int gx = tid & (tpr - 1); gx < gpr; gx += tpr) {
            uint32_t sx[4], tap[4], sel[4];
#pragma unroll
            for (int i = 0; i < 4; i++) { const uint2 c = colT[4 * gx + i]; sx[i] = c.x; tap[i] = c.y; }
            // k_resize's dword form on the LDS tile: the 4 outputs read source bytes sx[0] .. sx[0] + 7 of two rows (one 8-byte read each), v_perm_b32
            // spreads an output's two bytes into 16-bit halves, v_dot2_u32_u16 multiplies by the tap pair; wider spans take the byte form
            const bool span8 = sx[3] + 1u - sx[0] <= 7u;
#pragma unroll
            for (int i = 0; i < 4; i++) { const uint32_t k = sx[i] - sx[0]; sel[i] = k | 0x0c000c00u | ((k + 1u) << 16); }
            typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));
            for (int gy = tid >> tprLog; gy < rows; gy += rstep) {
                const uint2 t = rowT[gy];
                const uint8_t *r0 = A + (t.x & 0xFFFFu) * apitch, *r1 = A + (t.x >> 16) * apitch;
                const uint32_t bh0 = t.y << 16, bh1 = t.y & 0xFFFF0000u;       // vertical taps << 16: (b * x) >> 16 == mulhi(b << 16, x)
                uint32_t packed = 0;
                if (span8) {
                    const uint64_t s0 = reinterpret_cast<const U64 *>(r0 + sx[0])->v, s1 = reinterpret_cast<const U64 *>(r1 + sx[0])->v;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const uint32_t p0 = __builtin_amdgcn_perm((uint32_t)(s0 >> 32), (uint32_t)s0, sel[i]);
                        const uint32_t p1 = __builtin_amdgcn_perm((uint32_t)(s1 >> 32), (uint32_t)s1, sel[i]);
                        const uint32_t q0 = __builtin_amdgcn_udot2(__builtin_bit_cast(v2u16, p0), __builtin_bit_cast(v2u16, tap[i]), 0u, false);
                        const uint32_t q1 = __builtin_amdgcn_udot2(__builtin_bit_cast(v2u16, p1), __builtin_bit_cast(v2u16, tap[i]), 0u, false);
                        packed |= ((__umulhi(bh0, q0 >> 4) + __umulhi(bh1, q1 >> 4) + 2u) >> 2) << (8 * i);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const uint32_t a0 = tap[i] & 0xFFFFu, a1 = tap[i] >> 16;
                        const uint32_t q0 = r0[sx[i]] * a0 + (a1 ? r0[sx[i] + 1] * a1 : 0u), q1 = r1[sx[i]] * a0 + (a1 ? r1[sx[i] + 1] * a1 : 0u);
                        packed |= ((__umulhi(bh0, q0 >> 4) + __umulhi(bh1, q1 >> 4) + 2u) >> 2) << (8 * i);
                    }
                }
                *reinterpret_cast<uint32_t *>(B + gy * bw + 4 * gx) = packed;
                *reinterpret_cast<uint32_t *>(dbase + (long long)gy * D.pitch + 4 * gx) = packed;
            }
        }
        __syncthreads();
        uint8_t *t2 = A; A = B; B = t2;
        apitch = bw;
    }
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
#include <cstdlib>
__device__ __forceinline__ int min3i(int a, int b, int c) { return min(min(a, b), c); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }

// One WAVE per (frame, cell), four cells per 256-thread workgroup, no workgroup barrier anywhere: the wave stages its
// sub-image as dwords, tests 256 pixels per step, and emits in index order with a running offset.  LDS per wave is
// sized by the host from the largest cell of the current geometry (FastLds), so occupancy is not limited by LDS.
struct FastLds { int tp, sp, tileBytes, scBytes, maxIters, perWave; };

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- quick test and exact score, third formulation (round 3) --------------------------------------------------------------------
// Instruction classes on gfx950 (tools/valu_rate.hip, profiles/r02_valu_issue_rates.txt): plain add / sub / and / or / xor / right shift /
// mov issue in ~2.4 cycles per wave once two waves share a SIMD; every min / max, three-operand, packed, SDWA and DPP form takes ~4.2.
//
// Tile.  Column c of the LDS tile is image column iniX - 1 + c: the detection region (cv::FAST's 3-px margin inside the sub-image)
// ALWAYS starts at tile column 4, i.e. on a dword, whatever iniX is (the staging loads are unaligned 4-byte global loads).  A row of the
// region then is ceil(dw / 4) aligned 4-pixel groups with no partial first group (round 2 staged aligned dwords and lost up to one
// group per row to the shift).
//
// Quick test (necessary condition, per polarity): every arc of 9 contains 4 consecutive of the 8 EVEN circle positions, so a pixel can
// reach contrast T on the darker-ring side only if 4 consecutive even positions all have v - p_k >= T (brighter ring: p_k - v >= T).
// A lane takes the 4 pixels of one group; its operands come from 11 aligned dword reads of LDS.  Pixels (0, 2) and (1, 3) travel as
// 16-bit halves of two registers ("even" / "odd" pair).  Adding 2^b - T to the centre before ONE 32-bit subtraction of the packed ring
// pixels leaves "contrast >= T" in bit b of each half (the halves never borrow from each other: every half stays within 2^b +- 511).
// The four families (pair x polarity) use b = 12..15, one v_bfi_b32 each merges them into ONE word per circle position, and the
// "4 consecutive" rule (23 ANDs / ORs) runs once for all eight (pixel, polarity) combinations of the lane instead of four times.
// Ring entries = tile offset of the pixel | polarity << 15; a pixel's darker entry always precedes its brighter one.
constexpr int kRingCap = 640;        // linear: < 128 entries wait between steps, a step appends up to 512 (64 lanes x 4 pixels x 2 polarities)
constexpr int kScoredCap = 640;
// (ring pixels q in [0, 255] travel as 0x4100 + q: positive normal f16 bit patterns of one exponent, ordered like the integers)

__device__ __forceinline__ uint32_t pk_min3_f16(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t pk_max3_f16(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// inclusive prefix sum over the wave's 64 lanes: four row shifts and two row broadcasts (v_add_u32 with a DPP operand each)
__device__ __forceinline__ int wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);    // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);    // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);    // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);    // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);    // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);    // row_bcast:31 into rows 2 and 3
    return v;
}

// every bit position on its own: 4 consecutive of the 8 words have the bit set (circular)
__device__ __forceinline__ uint32_t four_consecutive(const uint32_t (&f)[8]) {
    uint32_t c[8];
#pragma unroll
    for (int k = 0; k < 8; k++) c[k] = f[k] & f[(k + 1) & 7];
    uint32_t any = c[0] & c[2];
#pragma unroll
    for (int k = 1; k < 8; k++) any |= c[k] & c[(k + 2) & 7];
    return any;
}
// (a & mask) | (b & ~mask) as ONE instruction.  Inline asm on purpose: written in C the compiler sees that only the masked bits are ever
// used, distributes the masks through the AND / OR network below and ends up with the four separate networks again.
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(mask), "v"(a), "v"(b));
    return r;
}

typedef __attribute__((address_space(3))) const uint8_t lds_cu8;
__device__ __forceinline__ uint32_t lds_addr(const uint8_t *p) { return (uint32_t)(uintptr_t)(lds_cu8 *)p; }

// The 16 circle pixels and the centre of ring entry 0 (7 x 7 neighbourhood with top-left corner t0) in the LOW halves of 17 registers, and
// those of entry 1 (t1) in the HIGH halves of 17 others: ds_read_u8 and ds_read_u8_d16_hi with immediate offsets (compile-time tile
// pitch).  The d16_hi form puts the byte where the packed arithmetic wants it, so that no shift is spent on packing -- but on a part
// with SRAM ECC (gfx950) a d16 load ZEROES the other half of its destination instead of keeping it, so the two entries cannot share a
// register at load time; one three-input bit operation per circle position merges them (and applies the polarity mask, see below).
// Each asm block carries its own s_waitcnt: the compiler does not track the LDS counter of inline asm, so no result may leave a block
// before it has arrived.
template <int CTP>
__device__ __forceinline__ void fast_ring_load(const uint8_t *t0g, const uint8_t *t1, int tp, uint32_t (&lo)[17], uint32_t (&hi)[17]) {
    if constexpr (CTP != 0) {
        lds_cu8 *t0 = (lds_cu8 *)t0g;
        const uint32_t A1 = lds_addr(t1);
#define RUMI_LD(reg, dx, dy) " %" #reg ", %17 offset:%18*(3+(" #dy "))+3+(" #dx ")\n\t"
#define RUMI_LD17(op)                                                                                              \
        op RUMI_LD(0, 0, 3)    op RUMI_LD(1, 1, 3)    op RUMI_LD(2, 2, 2)     op RUMI_LD(3, 3, 1)                      \
        op RUMI_LD(4, 3, 0)    op RUMI_LD(5, 3, -1)   op RUMI_LD(6, 2, -2)    op RUMI_LD(7, 1, -3)                     \
        op RUMI_LD(8, 0, -3)   op RUMI_LD(9, -1, -3)  op RUMI_LD(10, -2, -2)  op RUMI_LD(11, -3, -1)                   \
        op RUMI_LD(12, -3, 0)  op RUMI_LD(13, -3, 1)  op RUMI_LD(14, -2, 2)   op RUMI_LD(15, -1, 3)                    \
        op RUMI_LD(16, 0, 0)   "s_waitcnt lgkmcnt(0)"
        // entry 0: plain byte loads the compiler issues and tracks itself; they are queued BEFORE the asm block below (a volatile asm with a
        // memory clobber is not crossed), whose single s_waitcnt lgkmcnt(0) therefore covers all 34 loads in one LDS round trip
        asm("" : "+v"(t0));            // the base as one opaque register: all 17 offsets then are non-negative immediates of the load instruction
#define RUMI_LO(k, dx, dy) lo[k] = t0[(3 + (dy)) * CTP + 3 + (dx)];
        RUMI_LO(0, 0, 3)    RUMI_LO(1, 1, 3)    RUMI_LO(2, 2, 2)     RUMI_LO(3, 3, 1)
        RUMI_LO(4, 3, 0)    RUMI_LO(5, 3, -1)   RUMI_LO(6, 2, -2)    RUMI_LO(7, 1, -3)
        RUMI_LO(8, 0, -3)   RUMI_LO(9, -1, -3)  RUMI_LO(10, -2, -2)  RUMI_LO(11, -3, -1)
        RUMI_LO(12, -3, 0)  RUMI_LO(13, -3, 1)  RUMI_LO(14, -2, 2)   RUMI_LO(15, -1, 3)
        RUMI_LO(16, 0, 0)
#undef RUMI_LO
        asm volatile(RUMI_LD17("ds_read_u8_d16_hi")
                     : "=&v"(hi[0]), "=&v"(hi[1]), "=&v"(hi[2]), "=&v"(hi[3]), "=&v"(hi[4]), "=&v"(hi[5]), "=&v"(hi[6]), "=&v"(hi[7]), "=&v"(hi[8]),
                       "=&v"(hi[9]), "=&v"(hi[10]), "=&v"(hi[11]), "=&v"(hi[12]), "=&v"(hi[13]), "=&v"(hi[14]), "=&v"(hi[15]), "=&v"(hi[16])
                     : "v"(A1), "n"(CTP)
                     : "memory");
#undef RUMI_LD17
#undef RUMI_LD
    } else {
#define RUMI_RING(k, dx, dy) lo[k] = t0g[(3 + (dy)) * tp + 3 + (dx)]; hi[k] = (uint32_t)t1[(3 + (dy)) * tp + 3 + (dx)] << 16;
        RUMI_RING(0, 0, 3)    RUMI_RING(1, 1, 3)    RUMI_RING(2, 2, 2)     RUMI_RING(3, 3, 1)
        RUMI_RING(4, 3, 0)    RUMI_RING(5, 3, -1)   RUMI_RING(6, 2, -2)    RUMI_RING(7, 1, -3)
        RUMI_RING(8, 0, -3)   RUMI_RING(9, -1, -3)  RUMI_RING(10, -2, -2)  RUMI_RING(11, -3, -1)
        RUMI_RING(12, -3, 0)  RUMI_RING(13, -3, 1)  RUMI_RING(14, -2, 2)   RUMI_RING(15, -1, 3)
        RUMI_RING(16, 0, 0)
#undef RUMI_RING
    }
}

// exact scores of up to 128 ring entries, two per lane (entries `lane` and `lane + 64` of the batch: one LDS instruction then serves 64
// CONSECUTIVE entries, which lie within a few tile rows).
// A darker-ring entry is scored on COMPLEMENTED pixels (255 - p, 255 - v): its contrasts v - p_k are then the brighter-ring contrasts
// q_k - vq of the complemented data, so one network serves both polarities and the two entries of a lane may differ in polarity.  Per
// entry, with q_k = p_k ^ x (x = 0xFF darker, 0 brighter) and vq = v ^ x:  score = max over the 16 arcs of 9 of min q_k  -  vq  -  1.
// The XOR also sets the f16 exponent (0x4100) and rides on the instruction that merges the two entries' bytes; then 16 + 16 packed
// three-input minima and 8 maxima for both entries.
// A pixel cannot reach a positive score in both polarities (two arcs of 9 on a circle of 16 share two positions), so a hit stores its
// score byte unconditionally and is appended to the cell's SCORED LIST sl (tile offsets, ascending because the ring is filled in pixel
// order): NMS and emission then walk a few hundred listed pixels instead of the whole score map.  nScored counts all appends; once it
// passes kScoredCap the list is abandoned and the caller scans the map.
template <int CTP>
__device__ __forceinline__ void fast_score_batch(const uint8_t *tile, uint8_t *sc, uint16_t *sl, int &nScored, const uint16_t *ring, int n, int tp, int scDelta,
                                                 int tlow, int lane) {
    const int TP = CTP ? CTP : tp;
    const bool act0 = lane < n, act1 = lane + 64 < n;
    const uint32_t e0 = ring[lane], e1 = ring[lane + 64];
    const int a0 = act0 ? (int)(e0 & 0x7FFFu) : 3 * TP + 4, a1 = act1 ? (int)(e1 & 0x7FFFu) : 3 * TP + 4;
    // per half: 0x41FF for a darker-ring entry, 0x4100 for a brighter-ring one
    const uint32_t X = 0x41FF41FFu - ((e0 >> 15) | ((e1 >> 15) << 16)) * 0xFFu;
    uint32_t lo[17], hi[17], q[16];
    fast_ring_load<CTP>(tile + a0 - 3 * TP - 3, tile + a1 - 3 * TP - 3, TP, lo, hi);     // top-left corners of the 7 x 7 neighbourhoods: every offset is >= 0
#pragma unroll
    for (int k = 0; k < 16; k++) q[k] = (lo[k] | hi[k]) ^ X;                              // one v_bitop3_b32 each
    const uint32_t vc = lo[16] | hi[16];
    uint32_t lo3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) lo3[k] = pk_min3_f16(q[k], q[(k + 1) & 15], q[(k + 2) & 15]);
    uint32_t arc[16];
#pragma unroll
    for (int k = 0; k < 16; k++) arc[k] = pk_min3_f16(lo3[k], lo3[(k + 3) & 15], lo3[(k + 6) & 15]);
    uint32_t A = pk_max3_f16(arc[0], arc[1], arc[2]);
#pragma unroll
    for (int k = 3; k < 15; k += 2) A = pk_max3_f16(A, arc[k], arc[k + 1]);
    A = pk_max3_f16(A, arc[15], arc[15]);
    // per half: A = 0x4100 + max-min q, vq' = 0x4100 + vq; hit <=> A - vq' - 1 >= tlow.  D = (A | 0x8000) - (vq' + tlow + 1) stays within
    // 0x7E01 .. 0x80FE per half (no borrow between the halves) and carries the decision in bits 15 / 31; for a hit the low byte of
    // D + tlow is the score.
    const uint32_t D = (A | 0x80008000u) - ((vc ^ X) + (uint32_t)(tlow + 1) * 0x10001u);
    const uint32_t Sb = D + (uint32_t)tlow * 0x10001u;
    const bool hit0 = act0 && (D & 0x8000u) != 0, hit1 = act1 && (int32_t)D < 0;
    if (hit0) sc[a0 + scDelta] = (uint8_t)Sb;
    if (hit1) sc[a1 + scDelta] = (uint8_t)(Sb >> 16);
    const unsigned long long h0 = __ballot(hit0), h1 = __ballot(hit1);
    const int c0 = __popcll(h0), total = nScored + c0 + __popcll(h1);
    if (total <= kScoredCap) {                                       // wave-uniform: once the list has overflowed its content is never read
        const int pos0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(h0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)h0, nScored));
        const int pos1 = __builtin_amdgcn_mbcnt_hi((uint32_t)(h1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)h1, nScored + c0));   // ring order: entries 0..63, then 64..127
        if (hit0) sl[pos0] = (uint16_t)a0;
        if (hit1) sl[pos1] = (uint16_t)a1;
    }
    nScored = total;
}

// score map of one cell (CTP != 0: compile-time tile pitch; the score map shares the tile's pitch, so a pixel's score byte sits at
// its tile offset + scDelta).  Returns the number of scored-list appends.
template <int CTP>
__device__ __forceinline__ int fast_score_cell(const uint8_t *tile, uint8_t *sc, uint16_t *cl, uint16_t *sl, int tp, int dw, int dh, int tlow, int lane) {
    const int TP = CTP ? CTP : tp;
    const int ng = (dw + 3) >> 2;                         // aligned 4-pixel groups per row; the first starts at tile column 4
    const int nItems = ng * dh;
    const unsigned Mng = magic_of(ng);
    const int scDelta = -2 * TP - 3;                      // tile offset (py + 3) * TP + px + 4  ->  score byte (py + 1) * TP + px + 1
    // entry mask (2 bits per pixel) of a row's last group: pixels from column dw on lie outside the region
    const uint32_t mLast = 0xFFu >> (2 * (4 * ng - dw));
    const uint32_t T2 = (uint32_t)(tlow + 1);             // the contrast a circle pixel needs
    // flag bit of each (pair, polarity) family: even-pair darker 12, even-pair brighter 13, odd-pair darker 14, odd-pair brighter 15 -- bit
    // 2 i + polarity of (any >> 12) then belongs to pixel i of the group (pixel order 0, 1, 2, 3 = even.lo, odd.lo, even.hi, odd.hi)
    const uint32_t kDe = (0x1000u - T2) * 0x10001u, kBe = (0x2000u - T2) * 0x10001u, kDo = (0x4000u - T2) * 0x10001u, kBo = (0x8000u - T2) * 0x10001u;
    uint32_t *cl32 = reinterpret_cast<uint32_t *>(cl);
    int pending = 0, nScored = 0;
    for (int base = 0; base < nItems; base += 64) {
        const int ip = base + lane;
        const bool live = ip < nItems;
        const int row = live ? magic_div(ip, Mng) : 0, gi = live ? ip - mul24(row, ng) : 0;
        const int A = mul24(row + 3, TP) + 4 * (gi + 1);                      // tile offset of the group's first pixel
        const uint8_t *t = tile + A;
#define RUMI_DW(off) (*reinterpret_cast<const uint32_t *>(t + (off)))
        const uint32_t cM3 = RUMI_DW(-3 * TP), cP3 = RUMI_DW(3 * TP);
        const uint32_t l0 = RUMI_DW(-4), cc = RUMI_DW(0), r0 = RUMI_DW(4);
        const uint32_t lM2 = RUMI_DW(-2 * TP - 4), cM2 = RUMI_DW(-2 * TP), rM2 = RUMI_DW(-2 * TP + 4);
        const uint32_t lP2 = RUMI_DW(2 * TP - 4), cP2 = RUMI_DW(2 * TP), rP2 = RUMI_DW(2 * TP + 4);
#undef RUMI_DW
        // ring pixels of the (0, 2) pair ("e") and the (1, 3) pair ("o") as 16-bit halves, even circle positions in circular order:
        // (0,+3) (+2,+2) (+3,0) (+2,-2) (0,-3) (-2,-2) (-3,0) (-2,+2).  v_perm_b32 over {right | centre} or {centre | left} picks a shifted
        // pair in one instruction; the unshifted ones are an AND / shift + AND.
        uint32_t pe[8], po[8];
        pe[0] = cP3 & 0x00FF00FFu;                               po[0] = (cP3 >> 8) & 0x00FF00FFu;
        pe[1] = __builtin_amdgcn_perm(rP2, cP2, 0x0c040c02u);    po[1] = __builtin_amdgcn_perm(rP2, cP2, 0x0c050c03u);
        pe[2] = __builtin_amdgcn_perm(r0, cc, 0x0c050c03u);      po[2] = __builtin_amdgcn_perm(r0, cc, 0x0c060c04u);
        pe[3] = __builtin_amdgcn_perm(rM2, cM2, 0x0c040c02u);    po[3] = __builtin_amdgcn_perm(rM2, cM2, 0x0c050c03u);
        pe[4] = cM3 & 0x00FF00FFu;                               po[4] = (cM3 >> 8) & 0x00FF00FFu;
        pe[5] = __builtin_amdgcn_perm(cM2, lM2, 0x0c040c02u);    po[5] = __builtin_amdgcn_perm(cM2, lM2, 0x0c050c03u);
        pe[6] = __builtin_amdgcn_perm(cc, l0, 0x0c030c01u);      po[6] = __builtin_amdgcn_perm(cc, l0, 0x0c040c02u);
        pe[7] = __builtin_amdgcn_perm(cP2, lP2, 0x0c040c02u);    po[7] = __builtin_amdgcn_perm(cP2, lP2, 0x0c050c03u);
        const uint32_t ve = cc & 0x00FF00FFu, vo = (cc >> 8) & 0x00FF00FFu;
        const uint32_t cDe = ve + kDe, cBe = kBe - ve, cDo = vo + kDo, cBo = kBo - vo;
        uint32_t f[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
            f[k] = bfi(0x10001000u, cDe - pe[k], bfi(0x20002000u, pe[k] + cBe, bfi(0x40004000u, cDo - po[k], po[k] + cBo)));
        const uint32_t any = four_consecutive(f);
        // entry mask: bit 2 i + polarity for pixel i of the group
        uint32_t m = ((any >> 12) & 0xFu) | ((any >> 24) & 0xF0u);
        if (gi == ng - 1) m &= mLast;
        if (!live) m = 0;
        if (__ballot(m != 0) != 0) {
            // ring positions: entries of lower lanes first; within a lane pixel by pixel, darker before brighter
            const int cnt = __popc(m);
            const int incl = wave_incl_scan(cnt);
            uint16_t *w = cl + pending + incl - cnt;
            uint32_t e = (uint32_t)A;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (m & (1u << (2 * i))) *w++ = (uint16_t)e;
                if (m & (2u << (2 * i))) *w++ = (uint16_t)(e | 0x8000u);
                e++;
            }
            pending += __builtin_amdgcn_readlane(incl, 63);
            int head = 0;
            while (pending >= 128) {                       // a full batch: score it exactly
                wave_lds_fence();
                fast_score_batch<CTP>(tile, sc, sl, nScored, cl + head, 128, tp, scDelta, tlow, lane);
                head += 128;
                pending -= 128;
            }
            if (head) {                                    // the entries still waiting move to the front (fewer than 128, from beyond them)
                wave_lds_fence();
                const uint32_t q = cl32[(head >> 1) + lane];
                wave_lds_fence();
                if (2 * lane < pending) cl32[lane] = q;
            }
        }
    }
    wave_lds_fence();
    if (pending) fast_score_batch<CTP>(tile, sc, sl, nScored, cl, pending, tp, scDelta, tlow, lane);
    return nScored;
}

// One cell's place in its frame.
struct FastCell {
    const uint8_t *img;              // first staged byte: row iniY, column iniX - 1 (any alignment)
    long long cellIdx;
    int pitch, rows, cols, nd;       // image pitch; sub-image size; dwords per staged row
    int ox, oy;                      // cell origin relative to (16, 16): ci_j * wCell, ci_i * hCell
    bool live;
};
__device__ __forceinline__ FastCell fast_cell_geom(const DevParams *__restrict__ P, const ImgSrc &src, int cell, int frame, int32_t *__restrict__ cellCnt, int lane) {
    FastCell g;
    g.live = false;
    if (cell >= P->totalCells) return g;
    int level = 0;
    for (int l = 1; l < P->nlevels; l++)
        if (cell >= P->lv[l].cellBase) level = l;
    const DevLevel &L = P->lv[level];
    const int ci = cell - L.cellBase;
    const int ci_i = ci / L.nCols, ci_j = ci - ci_i * L.nCols;
    g.cellIdx = (long long)frame * P->totalCells + cell;
    const int iniY = kBorder + ci_i * L.hCell, iniX = kBorder + ci_j * L.wCell;
    const int maxY = min(iniY + L.hCell + 6, L.maxBY), maxX = min(iniX + L.wCell + 6, L.maxBX);
    g.cols = maxX - iniX; g.rows = maxY - iniY;
    // skip rules of ORBextractor.cc:752,760 and cv::FAST's 3-px margins
    if (iniY >= L.maxBY - 3 || iniX >= L.maxBX - 6 || g.cols < 7 || g.rows < 7) {
        if (lane == 0) cellCnt[g.cellIdx] = 0;
        return g;
    }
    // tile column c = image column iniX - 1 + c (iniX >= 16): the detection region starts at tile column 4.  A staged row is
    // ceil((cols - 6) / 4) + 2 dwords; its last byte is at most image column maxX + 3 <= width - 13, inside the row.
    g.img = level_base(src, P, level, frame, &g.pitch) + (long long)iniY * g.pitch + (iniX - 1);
    g.nd = ((g.cols - 6 + 3) >> 2) + 2;
    g.ox = ci_j * L.wCell; g.oy = ci_i * L.hCell;
    g.live = true;
    return g;
}
// Staging.  A lane owns ONE dword column c of the tile and one row r0 of every block of rps rows (rps = 64 / dword columns of the tile
// pitch): its offset into the sub-image and its LDS address are computed once, a block adds a wave-uniform row offset to both.  The
// last block is moved up so that it ends with the sub-image's last row (it re-loads a few rows of its predecessor): every lane of a
// block then is in range and no per-lane row test is needed.  The first kStageDepth blocks are in flight together (one memory round
// trip per cell for sub-images of up to kStageDepth x rps rows); lanes of columns beyond the cell's own width idle.  The loads are
// unaligned 4-byte GLOBAL accesses (the tile's column 0 is image column iniX - 1; buffer loads would drop the two low address bits).
constexpr int kStageDepth = 10;
template <int TPC>
__device__ __forceinline__ void fast_cell_stage(const FastCell &g, uint8_t *tile, int tp, int lane) {
    const int TP = TPC ? TPC : tp;
    const int ndT = TP >> 2, rps = min(64 / ndT, 7);           // a sub-image has at least 7 rows
    const int r0 = lane / ndT, c = lane - r0 * ndT;
    if (r0 < rps && c < g.nd) {
        const uint8_t *src = g.img + r0 * g.pitch + 4 * c;
        uint8_t *dst = tile + r0 * TP + 4 * c;
        const int nb = (g.rows + rps - 1) / rps, lastRow = g.rows - rps;
        uint32_t v[kStageDepth];
#pragma unroll
        for (int j = 0; j < kStageDepth; j++)
            if (j < nb) v[j] = reinterpret_cast<const U32 *>(src + (long long)min(j * rps, lastRow) * g.pitch)->v;
#pragma unroll
        for (int j = 0; j < kStageDepth; j++)
            if (j < nb) *reinterpret_cast<uint32_t *>(dst + min(j * rps, lastRow) * TP) = v[j];
        for (int j = kStageDepth; j < nb; j++)                    // taller sub-images: the rest, one round trip per block of rows
            *reinterpret_cast<uint32_t *>(dst + min(j * rps, lastRow) * TP) = reinterpret_cast<const U32 *>(src + (long long)min(j * rps, lastRow) * g.pitch)->v;
    }
}

// everything after the staging of one cell: score map, NMS, ordered emission
template <int TPC>
__device__ __forceinline__ void fast_cell_process(const DevParams *__restrict__ P, const FastLds &F, const FastCell &g, uint8_t *tile, uint8_t *sc,
                                                  uint32_t *__restrict__ cellBuf, int32_t *__restrict__ cellCnt, int lane) {
    const int TP = TPC ? TPC : F.tp;
    const int dw = g.cols - 6, dh = g.rows - 6;
    const unsigned Mdw = magic_of(dw), Mtp = magic_of(TP);
    for (int idx = lane * 16; idx < (dh + 2) * TP; idx += 1024) *reinterpret_cast<uint4 *>(&sc[idx]) = make_uint4(0, 0, 0, 0);   // (scBytes is a multiple of 16)
    wave_lds_fence();
    const int npx = dw * dh;
    const int scDelta = -2 * TP - 3;                                 // tile offset of a detection pixel -> its byte in the score map
    uint16_t *cl = reinterpret_cast<uint16_t *>(sc + F.scBytes);
    uint16_t *sl = cl + kRingCap;
    // Two passes, as upstream calls cv::FAST (:771-785): threshold iniThFAST first, and minThFAST only when the cell yields no key-point (after
    // NMS) at iniThFAST.  A pixel below the pass's threshold can neither be emitted nor suppress a neighbour (cv::FAST's score rows hold 0
    // for it, and NMS needs a strictly larger neighbour), so each pass scores only what reaches ITS threshold: at iniThFAST the quick test
    // passes a fraction of the pixels it passes at minThFAST, and textured cells never run the second pass.
    int thr = max(1, P->iniTh);
    uint32_t *out = cellBuf + g.cellIdx * P->maxCellCand;
    int found;
#pragma nounroll
    for (int pass = 0;; pass++) {
        const int nScored = fast_score_cell<TPC>(tile, sc, cl, sl, TP, dw, dh, thr, lane);
        wave_lds_fence();
        // NMS + emission in one sweep over the scored list (ascending pixel order = the row-major order cv::FAST emits in; every pixel at
        // most once); a cell with more than kScoredCap scored pixels scans its whole score map instead.  Two items per lane and sweep, all
        // their LDS reads issued together and the eight comparisons evaluated without short-circuit: a sweep costs two LDS round trips, not
        // ten.  Survivors go straight to the cell's output slots (a pass that finds nothing has written nothing).
        const bool listed = nScored <= kScoredCap;
        const int nItems = listed ? nScored : npx;
        found = 0;
        for (int base = 0; base < nItems; base += 128) {
            int si[2];
            bool in[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int k = base + 64 * h + lane;
                in[h] = k < nItems;
                if (listed) {
                    si[h] = (int)sl[in[h] ? k : 0] + scDelta;                     // score-map offset of the pixel
                } else {
                    const int kk = in[h] ? k : 0, py = magic_div(kk, Mdw);
                    si[h] = mul24(py + 1, TP) + (kk - mul24(py, dw)) + 1;
                }
            }
            int v[2];
            bool isMax[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint8_t *s = &sc[si[h]];
                v[h] = s[0];
                const int n0 = s[-TP - 1], n1 = s[-TP], n2 = s[-TP + 1], n3 = s[-1], n4 = s[1], n5 = s[TP - 1], n6 = s[TP], n7 = s[TP + 1];
                isMax[h] = in[h] & (v[h] > 0) & (v[h] > n0) & (v[h] > n1) & (v[h] > n2) & (v[h] > n3) & (v[h] > n4) & (v[h] > n5) & (v[h] > n6) & (v[h] > n7);
            }
            const unsigned long long b0 = __ballot(isMax[0]), b1 = __ballot(isMax[1]);
            const int c0 = __popcll(b0);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                if (isMax[h]) {
                    const unsigned long long b = h ? b1 : b0;
                    const int slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, found + (h ? c0 : 0)));
                    const int py1 = magic_div(si[h], Mtp), px1 = si[h] - mul24(py1, TP);     // score-map row / column = detection row / column + 1
                    const uint32_t x = (uint32_t)(px1 + 2 + g.ox), y = (uint32_t)(py1 + 2 + g.oy);
                    out[slot] = x | (y << 12) | ((uint32_t)v[h] << 24);
                }
            }
            found += c0 + __popcll(b1);
        }
        if (found > 0 || pass == 1) break;               // retry with minThFAST only if the first call found nothing (:783)
        thr = max(1, P->minTh);                                      // scores of the first pass that are still in the map are rewritten with the same values
    }
    if (lane == 0) cellCnt[g.cellIdx] = found;
}

// TPC: tile pitch (= score-map pitch) as a compile-time constant: the circle offsets and the NMS neighbours then are immediate LDS
// offsets instead of one address add each; 0 = run-time
// (bx, gx): the workgroup's column and the columns of the FAST part of the launch (the whole grid, or its first gx columns in the fused launch)
template <int TPC>
__device__ __forceinline__ void fast_cells_body(const DevParams *__restrict__ P, const ImgSrc &src, const FastLds &F, uint32_t *__restrict__ cellBuf,
                                                int32_t *__restrict__ cellCnt, unsigned bx, unsigned gx) {
    extern __shared__ __attribute__((aligned(16))) uint8_t fl[];
    // the wave index as a scalar: everything that depends only on the cell (geometry, magic numbers, LDS bases) then runs on the scalar unit
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned wg = xcd_swizzle(blockIdx.y * gx + bx, gx * gridDim.y);
    const int wpg = blockDim.x >> 6;
    const int cell = (wg % gx) * wpg + wave, frame = wg / gx;
    uint8_t *tile = fl + (size_t)wave * F.perWave;
    uint8_t *sc = tile + F.tileBytes;
    const int TP = TPC ? TPC : F.tp;
    const FastCell gA = fast_cell_geom(P, src, cell, frame, cellCnt, lane);
    if (!gA.live) return;
    fast_cell_stage<TPC>(gA, tile, TP, lane);
    fast_cell_process<TPC>(P, F, gA, tile, sc, cellBuf, cellCnt, lane);
}
template <int TPC>
__global__ __launch_bounds__(256) void k_fast_cells(const DevParams *__restrict__ P, ImgSrc src, FastLds F,
                                                    uint32_t *__restrict__ cellBuf, int32_t *__restrict__ cellCnt) {
    fast_cells_body<TPC>(P, src, F, cellBuf, cellCnt, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------------------------
// Candidate compaction: one workgroup per frame concatenates the cell lists in cell order (levels
// ascending, cells row-major) into cand[frame][...] and writes levelStart[frame][0..nlevels].
// A level that would exceed its capacity is truncated and flagged (bit 4 of the call's error word); the host turns that into
// RUMI_E_CAPACITY.
// ------------------------------------------------------------------------------------------------
// 256 threads (a 1024-thread workgroup waits for a CU with sixteen free wave slots beside the other streams' kernels: 0.42 ms per 256-frame launch
// in the pipelined step against 0.03 ms alone -- without costing the step anything measurable; one frame: 6.7 -> ~3 us)
constexpr int kCompactThreads = 256;
__global__ __launch_bounds__(kCompactThreads) void k_compact(const DevParams *__restrict__ P, const uint32_t *__restrict__ cellBuf,
                                                 const int32_t *__restrict__ cellCnt, uint32_t *__restrict__ cand,
                                                 int32_t *__restrict__ levelStart, int32_t *__restrict__ errFlag) {
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
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
        if (sStart[c1] - sStart[c0] > P->lv[tid].candCap) atomicOr(errFlag, 16);
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
// pass to u32, (v + 32768) >> 16, BORDER_REFLECT_101 at the level's own edges.
// ------------------------------------------------------------------------------------------------
// Register formulation: a lane owns a 4-pixel column strip, a wave walks kBlurRows output rows top to bottom.
// Per source row: ONE aligned dword load per lane; the left / right neighbours' dwords arrive by DPP shuffles (the two
// outer lanes load their halo dwords); the 7-tap row pass runs on the 10 unpacked bytes, the column pass on a 7-deep
// register ring of row results; 4 output pixels leave as one dword store.  No LDS, no barriers.
// Edges (no border is stored around a level): rows above / below the level are the mirrored rows (a row index, wave-uniform); the three
// columns left of column 0 are bytes 3, 2, 1 of the first dword (one v_perm_b32 in the first strip block); columns from w on are mirrored
// bytes fetched by the few lanes whose dword touches them (byte loads of the same cache lines, only in waves that hold the right edge).
// output rows a wave walks: 64 for batches (6 halo rows per 64: +1.3 % on the pipelined step over 32, which was +1.7 % over 16), 16 for a few
// frames (a wave's walk is a chain of dependent row loads and one frame fills few waves: the device chain of a one-frame call takes 107-110 us
// with 16 rows, 116-126 with 32, 103-117 with 8 or 4 on the same box)
constexpr int kBlurRowsSmall = 16, kBlurRowsBatch = 64;

// all levels in one launch: workgroup `lin` of a frame belongs to the level whose [base, base + gx * gy) range holds it
struct BlurGrid { int base[kMaxLevels + 1]; int gx[kMaxLevels]; int bw[kMaxLevels]; };   // bw: pixels a wave's strips cover (256, or less: see launch_blur)
// a * b + c on 24-bit operands as ONE v_mad_u32_u24 (the compiler splits the C expression into a multiply and a 3-input add)
__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// VARIANT: RumiOrbConfig.blur_variant -- 0: taps {18,34,48,56,..}/256 of the fixed-point GaussianBlur of OpenCV >= 3.4.2; 1: the integer-scaled float
// kernel {18,34,49,55,..}/256 of 3.4.0 / 3.4.1 (sum 257: the result is saturated)
template <int VARIANT, int kBlurRows>
__device__ __forceinline__ void blur_body(const DevParams *__restrict__ P, const ImgSrc &src, const BlurGrid &G, unsigned bxg, unsigned gxg) {
    constexpr uint32_t kT2 = VARIANT ? 49u : 48u, kT3 = VARIANT ? 55u : 56u;      // taps at distance 1 and 0 (18 and 34 are common)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (scalar: the row walk is scalar arithmetic)
    const unsigned wg = xcd_swizzle(blockIdx.y * gxg + bxg, gxg * gridDim.y);
    const int frame = wg / gxg, lin = wg % gxg;
    int level = 0;
    for (int l = 1; l < P->nlevels; l++)
        if (lin >= G.base[l]) level = l;
    const DevLevel &L = P->lv[level];
    const int bx = (lin - G.base[level]) % G.gx[level], by = (lin - G.base[level]) / G.gx[level];
    const int bw = G.bw[level];
    const int xa = bx * bw + lane * 4;                           // first pixel of my strip (lanes from bw / 4 on only feed their left neighbour's halo)
    const int y0 = (by * 4 + wave) * kBlurRows;
    if (y0 >= L.h) return;                                       // whole wave (wave-uniform)
    int pitch;
    const uint8_t *img = level_base(src, P, level, frame, &pitch);
    uint8_t *out = src.blur + (long long)frame * P->arenaStride + L.off;
    const int w = L.w, h = L.h;
    // Right edge.  The dword that holds column w - 1 may be partial and the one after it lies wholly beyond the row, yet both feed the
    // halos of the last strips: their missing bytes are the mirrored columns 2 (w - 1) - x, which sit in the same lane or one / two lanes to
    // the left.  In the wave that holds the edge every lane rebuilds its dword from {own, left, left-left} with two v_perm_b32 whose
    // selectors are fixed per lane (identity away from the edge).  The host picks the strip width of a level (G.bw) so that the partial dword
    // is never lane 0 or 1 of a wave and a wave's last producing lane never needs a halo dword from beyond the row edge out of memory.
    const int xLast = (w - 1) & ~3;                              // last dword that holds a pixel of the row
    const int xl = min(xa, xLast);
    const bool firstBlock = bx == 0;
    const bool edgeWave = bx * bw + bw + 4 > w;                  // a dword of this wave (its right halo included) reaches column w or beyond (wave-uniform)
    const bool produce = lane * 4 < bw && xa < w;
    uint32_t selA = 0x03020100u, selB = 0x07060504u;             // identity: keep my own four bytes
    if (edgeWave && xa + 3 >= w && xa <= xLast + 4) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int x = xa + i;
            if (x < w) continue;
            const int sx = 2 * (w - 1) - x, d = (xa - (sx & ~3)) >> 2;                 // mirrored column, lanes to the left (0, 1 or 2 for every byte that is used)
            const uint32_t a = (d == 1 ? 4u : 0u) + (uint32_t)(sx & 3);                 // byte of {left-left (0-3), left (4-7)}
            const uint32_t b = d == 0 ? 4u + (uint32_t)(sx & 3) : (uint32_t)i;          // byte of {gathered (0-3), own (4-7)}
            selA = (selA & ~(0xFFu << (8 * i))) | (a << (8 * i));
            selB = (selB & ~(0xFFu << (8 * i))) | (b << (8 * i));
        }
    }
    int ring[7][4];                                              // row results of the last seven source rows; slot = source row mod 7 of this walk
#pragma unroll
    for (int k = 0; k < 7; k++)
#pragma unroll
        for (int i = 0; i < 4; i++) ring[k][i] = 0;
    const int yEnd = min(y0 + kBlurRows, h), rEnd = yEnd + 3;
    // the walk is unrolled by seven so that the ring never moves: source row r0 + j lands in slot j, and the taps of the output row it
    // completes sit at compile-time slots (a runtime ring costs 24 register moves per row)
    uint8_t *orow = out + (long long)(y0 - 6) * L.pitch + xa - L.pitch;
    // the halo dword of the wave's outer lanes: lane 0 reads the dword left of its own (but in the first strip block, where it is the mirrored
    // bytes of its own), lane 63 of a 256-pixel wave the one to the right
    const int haloOff = lane == 0 ? (firstBlock ? 0 : -4) : (lane == 63 && !edgeWave && bw == 256 ? 4 : 0);
    // the source rows of the NEXT seven are fetched while the current seven are filtered (a wave's walk is otherwise a chain of
    // load -> filter -> load; rows past the walk's end re-read its last row).  256 frames alone on the device: 260 -> 192 us at 83 registers
    // (5 waves a SIMD); forced to 80 registers / 6 waves (one spill) 217 us, held at 4 waves 207 us, two register sets taking turns 88 registers
    uint32_t Cn[7], Hn[7];
    auto fetch = [&](int r, uint32_t &C, uint32_t &H) {
        const int rc = min(r, rEnd - 1);
        const int rr = rc < 0 ? -rc : (rc >= h ? 2 * (h - 1) - rc : rc);        // rows -3..-1 and h..h+2 mirror into the level
        const uint8_t *row = img + (long long)rr * pitch + xl;
        C = *reinterpret_cast<const uint32_t *>(row);
        H = 0;
        if (haloOff) H = *reinterpret_cast<const uint32_t *>(row + haloOff);
    };
#pragma unroll
    for (int j = 0; j < 7; j++) fetch(y0 - 3 + j, Cn[j], Hn[j]);
    uint32_t Cm[7], Hm[7];                                       // the seven being filtered
    auto walk7 = [&](const uint32_t (&Cc)[7], const uint32_t (&Hc)[7], int r0) {
#pragma unroll
        for (int j = 0; j < 7; j++) {
            const int r = r0 + j;
            if (r >= rEnd) break;                                // wave-uniform
            orow += L.pitch;
            uint32_t C = Cc[j];
            if (edgeWave) {
                const uint32_t c1 = __shfl_up(C, 1), c2 = __shfl_up(c1, 1);
                C = __builtin_amdgcn_perm(C, __builtin_amdgcn_perm(c1, c2, selA), selB);
            }
            uint32_t Lw = __shfl_up(C, 1), Rw = __shfl_down(C, 1);
            if (lane == 0) Lw = firstBlock ? __builtin_amdgcn_perm(C, C, 0x01020300u) : Hc[j];
            if (lane == 63 && !edgeWave && bw == 256) Rw = Hc[j];   // (only a 256-pixel wave has a producing lane 63)
            // row pass on packed bytes: output i needs the 7 bytes S[i+1 .. i+7] of the 12-byte run {Lw, C, Rw}; two byte-dot-products
            // (v_dot4_u32_u8) against the taps {18,34,48,56} and {48,34,18,0} give the exact integer sum (<= 65 280)
            constexpr uint32_t tA = 18u | (34u << 8) | (kT2 << 16) | (kT3 << 24), tB = kT2 | (34u << 8) | (18u << 16);
            const uint32_t A0 = __builtin_amdgcn_alignbyte(C, Lw, 1), A1 = __builtin_amdgcn_alignbyte(C, Lw, 2), A2 = __builtin_amdgcn_alignbyte(C, Lw, 3);
            const uint32_t B0 = __builtin_amdgcn_alignbyte(Rw, C, 1), B1 = __builtin_amdgcn_alignbyte(Rw, C, 2), B2 = __builtin_amdgcn_alignbyte(Rw, C, 3);
            ring[j][0] = (int)__builtin_amdgcn_udot4(B0, tB, __builtin_amdgcn_udot4(A0, tA, 0u, false), false);
            ring[j][1] = (int)__builtin_amdgcn_udot4(B1, tB, __builtin_amdgcn_udot4(A1, tA, 0u, false), false);
            ring[j][2] = (int)__builtin_amdgcn_udot4(B2, tB, __builtin_amdgcn_udot4(A2, tA, 0u, false), false);
            ring[j][3] = (int)__builtin_amdgcn_udot4(Rw, tB, __builtin_amdgcn_udot4(C, tA, 0u, false), false);
            const int y = r - 3;                                 // slots (j+1)%7 .. j now hold rows y-3 .. y+3
            if (y >= y0 && produce) {
                uint32_t o[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    // rounding constant folded into the first multiply-add; the result's byte 2 is the output pixel (sum <= 255 * 65536 + 32768)
                    // row sums are <= 65 280 and their pairs <= 130 560: 24-bit multiply-adds (v_mad_u32_u24: tap and accumulation in one instruction)
                    uint32_t acc = mad_u24(kT3, (uint32_t)ring[(j + 4) % 7][i], 32768u);
                    acc = mad_u24(kT2, (uint32_t)(ring[(j + 3) % 7][i] + ring[(j + 5) % 7][i]), acc);
                    acc = mad_u24(34u, (uint32_t)(ring[(j + 2) % 7][i] + ring[(j + 6) % 7][i]), acc);
                    acc = mad_u24(18u, (uint32_t)(ring[(j + 1) % 7][i] + ring[j][i]), acc);
                    if (VARIANT) acc = min(acc, 0x00FFFFFFu);            // taps sum to 257: saturate_cast<uchar>
                    o[i] = acc;
                }
                // byte 2 of the four sums -> one dword (v_perm_b32: selectors 0-3 take from the second operand, 4-7 from the first, 0x0c = zero);
                // the blurred arena's rows are padded to 64 B, so a whole dword always fits in the row
                const uint32_t p01 = __builtin_amdgcn_perm(o[1], o[0], 0x0c0c0602u), p23 = __builtin_amdgcn_perm(o[3], o[2], 0x06020c0cu);
                *reinterpret_cast<uint32_t *>(orow) = p01 | p23;         // orow = out + y * pitch + xa
            }
        }
    };
    for (int r0 = y0 - 3; r0 < rEnd; r0 += 7) {
#pragma unroll
        for (int j = 0; j < 7; j++) Cm[j] = Cn[j], Hm[j] = Hn[j];
        if (r0 + 7 < rEnd) {
#pragma unroll
            for (int j = 0; j < 7; j++) fetch(r0 + 7 + j, Cn[j], Hn[j]);
        }
        walk7(Cm, Hm, r0);
    }
}
template <int VARIANT, int kBlurRows>
__global__ __launch_bounds__(256) void k_blur(const DevParams *__restrict__ P, ImgSrc src, BlurGrid G) {
    blur_body<VARIANT, kBlurRows>(P, src, G, blockIdx.x, gridDim.x);
}
// A few frames (the Tracking thread's call): FAST and the blur in ONE launch, the first gxFast workgroup columns FAST cells, the rest blur strips.
// Both only read the pyramid; as two launches the blur goes to a side stream, and the event that forks it stalls the main queue for ~20 us on
// this runtime (and the join for ~5): more than the blur takes.
template <int TPC, int VARIANT>
__global__ __launch_bounds__(256) void k_fast_blur(const DevParams *__restrict__ P, ImgSrc src, FastLds F, uint32_t *__restrict__ cellBuf,
                                                   int32_t *__restrict__ cellCnt, BlurGrid G, unsigned gxFast) {
    if (blockIdx.x < gxFast) fast_cells_body<TPC>(P, src, F, cellBuf, cellCnt, blockIdx.x, gxFast);
    else blur_body<VARIANT, kBlurRowsSmall>(P, src, G, blockIdx.x - gxFast, gridDim.x - gxFast);
}

// ------------------------------------------------------------------------------------------------
// Orientation + descriptor + output assembly: one HALF wave (32 lanes) per selected key-point, eight key-points per workgroup.
//   The arithmetic that is the same for every lane of a key-point (fastAtan2, the libm sinf / cosf restatement in double precision, the
//   record) is a third of the kernel: with two key-points per wave an instruction serves both.
//   IC_Angle: integer moments over the radius-15 disc of the UN-blurred level, lane = disc column;
//   rBRIEF:   lane l evaluates test pairs l, l+32, ... l+224; __ballot packs 32 bits per key-point at a time, which
//             is exactly the descriptor's little-endian bit order (bit k of byte i = pair 8i+k).
// ------------------------------------------------------------------------------------------------
// sum over the 32 lanes of a half wave, returned in every lane of that half: DPP adds inside the rows of 16 (the row's total lands in its
// lane 15), row_bcast:15 carries it into the next row, lanes 31 / 63 then hold the two totals (five ds_bpermute round trips otherwise)
__device__ __forceinline__ int half_wave_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);          // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);          // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);          // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);          // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);          // row_bcast:15 -> rows 1 and 3
    const int lo = __builtin_amdgcn_readlane(v, 31), hi = __builtin_amdgcn_readlane(v, 63);
    return (threadIdx.x & 32) ? hi : lo;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

constexpr int kDiscP = 48, kPatchP = 48;       // LDS row pitches: 36 and 40 staged bytes per row (31 / 37 + alignment slack), rows 16-byte aligned for b128 stores
constexpr int kKpPerWg = 8;                    // half waves of a workgroup
// kKpGroups: key-points a half wave handles one after the other (the next one's pixels are in flight meanwhile).  Two for batches: with four,
// the workgroups resident on an XCD span five frames instead of two and a half, their pyramids no longer fit its L2 and the kernel fetches
// 1.7x the bytes (FETCH_SIZE).  One for a handful of frames: there are not enough workgroups to fill the chip otherwise.

// kAssemble (calls of a few frames): k_assemble's work -- concatenate the levels, the lapping rule's slots (ORBextractor.cc:1077-1085), the frame's
// {n, monoIndex} -- is done by every workgroup for its own key-points in its prologue (a count over the frame's <= ~1100 selected key-points: four
// loads a thread), so that launch and its ~7 us on the dependent chain of a one-frame call disappear; workgroup 0 of a frame writes the counts and,
// for calls whose results go straight to pinned host memory, the call's final error word.
struct AssembleArgs {
    const uint32_t *selLevel; const int32_t *selLevelCnt; int selLevelCap, lap0, lap1;
    int32_t *counts; long long countsStride; int32_t *errFlag, *errMirror;
    uint32_t *selPackedOut, *selMetaOut; int32_t *selCountOut;      // k_assemble's arrays are still written (the parity taps read them)
};
template <int kKpGroups, bool kAssemble>
__global__ __launch_bounds__(256, 7) void k_orient_desc(const DevParams *__restrict__ P, ImgSrc src,
                                                     const uint32_t *__restrict__ selPacked,
                                                     const uint32_t *__restrict__ selMeta,
                                                     const int32_t *__restrict__ selCount, int selCap,
                                                     RumiKeyPoint *__restrict__ kpOut, long long kpStride, uint8_t *__restrict__ descOut,
                                                     long long descStride, int outCap, AssembleArgs A) {
    // per key-point: the 31-row disc neighbourhood of the un-blurred level, THEN (in the same LDS: the moments are done with the disc before the
    // descriptor wants the patch) the 37-row patch of the blurred level, staged by the half wave that owns the key-point and read by nobody else:
    // no workgroup barrier anywhere past the pattern table's.  18 KB per workgroup: seven workgroups per CU (30 KB with both resident: five)
    static_assert(kDiscP == kPatchP, "the disc and the patch share their rows");
    __shared__ __attribute__((aligned(16))) uint8_t sWin[kKpPerWg][37 * kPatchP];
    __shared__ __attribute__((aligned(16))) float sPat[256 * 4];
    __shared__ int4 sLv[kMaxLevels];              // per level: offset and pitch of the un-blurred image (level 0 = the caller's frame), of the blurred one
    __shared__ float2 sLvF[kMaxLevels];           // scale, patch size
    reinterpret_cast<float4 *>(sPat)[threadIdx.x] = reinterpret_cast<const float4 *>(c_patternF.v)[threadIdx.x];
    if (threadIdx.x < (unsigned)P->nlevels) {
        const DevLevel &Lv = P->lv[threadIdx.x];
        sLv[threadIdx.x] = threadIdx.x == 0 ? make_int4(0, src.l0Pitch, (int)Lv.off, Lv.pitch) : make_int4((int)Lv.off, Lv.pitch, (int)Lv.off, Lv.pitch);
        sLvF[threadIdx.x] = make_float2(Lv.scale, Lv.patchSize);
    }
    const int lane = threadIdx.x & 31, hw = threadIdx.x >> 5;             // lane within the half wave, half-wave index 0..7
    const unsigned wg = xcd_swizzle(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);   // a frame's key-points share one L2
    const int kb = (wg % gridDim.x) * (kKpPerWg * kKpGroups) + hw, frame = wg / gridDim.x;
    int cnt;
    __shared__ uint32_t sOwnPk[kAssemble ? kKpPerWg * kKpGroups : 1], sOwnMt[kAssemble ? kKpPerWg * kKpGroups : 1];
    if constexpr (!kAssemble) {
        cnt = selCount[frame];
        __syncthreads();
    } else {
        __shared__ int sLvStart[kMaxLevels + 1], sRed[2], sOwnF[kKpPerWg * kKpGroups];
        const int nl = P->nlevels, tid = threadIdx.x;
        if (tid == 0) {
            int run = 0;
            for (int l = 0; l < nl; l++) { sLvStart[l] = run; run += A.selLevelCnt[(long long)frame * nl + l]; }
            sLvStart[nl] = run; sRed[0] = 0; sRed[1] = 0;
        }
        __syncthreads();
        const int total = sLvStart[nl], kbase = (wg % gridDim.x) * (kKpPerWg * kKpGroups);
        const bool first = wg % gridDim.x == 0, over = total > selCap;
        int32_t *counts = reinterpret_cast<int32_t *>(reinterpret_cast<uint8_t *>(A.counts) + frame * A.countsStride);
        if (over) {                                           // k_assemble's refusal: more key-points than the selection arrays hold
            if (first && tid == 0) { A.selCountOut[frame] = 0; counts[0] = total; counts[1] = 0; const int old = atomicOr(A.errFlag, 8); if (A.errMirror) *A.errMirror = old | 8; }
            return;
        }
        cnt = total;
        auto key_at = [&](int k, int *levelOut) -> uint32_t {
            int level = 0;
            while (k >= sLvStart[level + 1]) level++;
            *levelOut = level;
            return A.selLevel[((long long)frame * nl + level) * A.selLevelCap + (k - sLvStart[level])];
        };
        auto lapped = [&](uint32_t pk, int level) -> bool {
            float x = (float)((int)(pk & 0xFFF) + kBorder);
            if (level != 0) x = x * P->lv[level].scale;
            return x >= (float)A.lap0 && x <= (float)A.lap1;
        };
        int before = 0, all = 0;
        for (int k = tid; k < total; k += 256) {
            int level;
            const uint32_t pk = key_at(k, &level);
            const int f = lapped(pk, level) ? 1 : 0;
            all += f; before += k < kbase ? f : 0;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { before += __shfl_xor(before, o); all += __shfl_xor(all, o); }
        if ((tid & 63) == 0) { atomicAdd(&sRed[0], before); atomicAdd(&sRed[1], all); }
        if (tid < kKpPerWg * kKpGroups) {
            const int k = kbase + tid;
            int level = 0;
            uint32_t pk = 0;
            int f = 0;
            if (k < total) { pk = key_at(k, &level); f = lapped(pk, level) ? 1 : 0; }
            sOwnPk[tid] = pk; sOwnMt[tid] = (uint32_t)level; sOwnF[tid] = f;
        }
        __syncthreads();
        if (tid == 0) {
            int b = sRed[0];
            for (int j = 0; j < kKpPerWg * kKpGroups; j++) {
                const int k = kbase + j;
                if (k >= total) break;
                const int f = sOwnF[j], slot = f ? (total - 1 - b) : (k - b);
                b += f;
                sOwnMt[j] |= (uint32_t)slot << 8;
            }
            if (first) {
                counts[0] = total; counts[1] = total - sRed[1];      // {n, monoIndex}
                A.selCountOut[frame] = total;
                if (A.errMirror) *A.errMirror = *A.errFlag;
            }
        }
        __syncthreads();
        if (tid < kKpPerWg * kKpGroups && kbase + tid < total) {
            A.selPackedOut[(long long)frame * selCap + kbase + tid] = sOwnPk[tid];
            A.selMetaOut[(long long)frame * selCap + kbase + tid] = sOwnMt[tid];
        }
    }
#ifdef RUMI_OD_STAMP
    long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stLast = clock64();
#define OD_STAMP(k) do { const long long t_ = clock64(); st[k] += t_ - stLast; stLast = t_; } while (0)
#else
#define OD_STAMP(k) do { } while (0)
#endif

    struct __attribute__((packed, aligned(4))) Q16 { uint32_t x, y, z, w; };      // (dword-aligned wide loads)
    struct __attribute__((packed, aligned(4))) Q8 { uint32_t x, y; };
    auto ld16 = [](const uint8_t *q) { const Q16 t = *reinterpret_cast<const Q16 *>(q); return make_uint4(t.x, t.y, t.z, t.w); };
    auto ld8 = [](const uint8_t *q) { const Q8 t = *reinterpret_cast<const Q8 *>(q); return make_uint2(t.x, t.y); };
    struct Staged { uint4 d0, d1, p0, p1, q0, q1; uint2 p2, q2; uint32_t d2; };
    const bool dRow = lane < 31, qRow = lane < 5;
    const uint8_t *frame0 = src.l0 + (long long)frame * src.l0FrameStride, *framePyr = src.pyr + (long long)frame * P->arenaStride,
                  *frameBlur = src.blur + (long long)frame * P->arenaStride;
    // lane = row: a row's 36 / 40 bytes are two 16-byte loads and a 4- / 8-byte one (dword-aligned addresses; the 37 rows of the patch take a
    // second, five-lane trip); no index arithmetic
    auto fetch_disc = [&](uint32_t pk, uint32_t meta, bool live, Staged &S) {
        if (!live || !dRow) return;
        const int level = meta & 0xFF, x = (int)(pk & 0xFFF) + kBorder, y = (int)((pk >> 12) & 0xFFF) + kBorder;
        const int4 lv = sLv[level];                                       // (a frame's arena is far below 2 GB: 32-bit offsets)
        const uint8_t *cr = (level == 0 ? frame0 : framePyr) + (lv.x + (y - kHalfPatch + lane) * lv.y + ((x - kHalfPatch) & ~3));
        S.d0 = ld16(cr); S.d1 = ld16(cr + 16); S.d2 = *reinterpret_cast<const uint32_t *>(cr + 32);
    };
    auto fetch_patch = [&](uint32_t pk, uint32_t meta, bool live, Staged &S) {
        if (!live) return;
        const int level = meta & 0xFF, x = (int)(pk & 0xFFF) + kBorder, y = (int)((pk >> 12) & 0xFFF) + kBorder;
        const int4 lv = sLv[level];
        const uint8_t *br = frameBlur + (lv.z + (y - 18 + lane) * lv.w + ((x - 18) & ~3)), *br2 = br + 32 * lv.w;
        S.p0 = ld16(br); S.p1 = ld16(br + 16); S.p2 = ld8(br + 32);
        if (qRow) { S.q0 = ld16(br2); S.q1 = ld16(br2 + 16); S.q2 = ld8(br2 + 32); }
    };
    auto stage_disc = [&](bool live, const Staged &S) {
        if (!live || !dRow) return;
        uint8_t *dst = &sWin[hw][lane * kDiscP];
        *reinterpret_cast<uint4 *>(dst) = S.d0; *reinterpret_cast<uint4 *>(dst + 16) = S.d1; *reinterpret_cast<uint32_t *>(dst + 32) = S.d2;
    };
    auto stage_patch = [&](bool live, const Staged &S) {
        if (!live) return;
        {
            uint8_t *dst = &sWin[hw][lane * kPatchP];
            *reinterpret_cast<uint4 *>(dst) = S.p0; *reinterpret_cast<uint4 *>(dst + 16) = S.p1; *reinterpret_cast<uint2 *>(dst + 32) = S.p2;
        }
        if (qRow) {
            uint8_t *dst = &sWin[hw][(lane + 32) * kPatchP];
            *reinterpret_cast<uint4 *>(dst) = S.q0; *reinterpret_cast<uint4 *>(dst + 16) = S.q1; *reinterpret_cast<uint2 *>(dst + 32) = S.q2;
        }
    };
    auto orientation = [&](uint32_t pk) -> float {
        const int x = (int)(pk & 0xFFF) + kBorder;
        const int xd = (x - kHalfPatch) & ~3;
        // IC_Angle (ORBextractor.cc:73-97): lane = column u of the disc; the disc is symmetric (|u| <= umax[|v|]  <=>  |v| <= umax[|u|]), so a
        // lane's rows are |v| <= umax[|u|], known before the loop; m10 = u * (sum of the column), m01 = sum of v * pixel
        const uint8_t *dc = &sWin[hw][kHalfPatch * kDiscP + (x - xd)];
        const int u = lane - kHalfPatch;
        const int vmaxU = lane < 31 ? P->umax[u < 0 ? -u : u] : -1;
        // rows +v and -v share their bound: one compare masks both; every row of the staged disc exists, so the reads are unconditional
        const int mid = dc[u];
        int colSum = vmaxU >= 0 ? mid : 0, m01 = 0;
#pragma unroll
        for (int v = 1; v <= kHalfPatch; v++) {
            const int lo = dc[-v * kDiscP + u], hi = dc[v * kDiscP + u];
            const bool in = v <= vmaxU;
            colSum += in ? lo + hi : 0;
            m01 += in ? v * (hi - lo) : 0;
        }
        int m10 = u * colSum;
        m10 = half_wave_sum(m10);
        m01 = half_wave_sum(m01);
        OD_STAMP(4);
        return fast_atan2_deg((float)m01, (float)m10);
    };
    auto describe = [&](uint32_t pk, uint32_t meta, float angle) {
        const int level = meta & 0xFF, slot = (int)(meta >> 8);
        const int x = (int)(pk & 0xFFF) + kBorder, y = (int)((pk >> 12) & 0xFFF) + kBorder, score = (int)(pk >> 24);
        const int xp = (x - 18) & ~3;
        // computeOrbDescriptor (ORBextractor.cc:99-143) on the blurred level
        const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
        const float ang = angle * factorPI;
        const float a = cosf_glibc(ang), b = sinf_glibc(ang);
        const uint8_t *bc = &sWin[hw][18 * kPatchP + (x - xp)];
        uint32_t w = 0;                                                   // lane j of the half wave ends up with descriptor word j
        OD_STAMP(5);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float4 pt = reinterpret_cast<const float4 *>(sPat)[j * 32 + lane];
            const float x0 = pt.x, y0 = pt.y, x1 = pt.z, y1 = pt.w;
            const int r0 = cv_round_f(x0 * b + y0 * a), c0 = cv_round_f(x0 * a - y0 * b);
            const int r1 = cv_round_f(x1 * b + y1 * a), c1 = cv_round_f(x1 * a - y1 * b);
            const int t0 = bc[r0 * kPatchP + c0], t1 = bc[r1 * kPatchP + c1];
            const unsigned long long bal = __ballot(t0 < t1);             // both key-points of the wave; lanes j and 32 + j keep their halves
            const uint32_t lo = (uint32_t)bal, hi = (uint32_t)(bal >> 32);
            // (v_writelane reads its scalar operand early: the compare that wrote it needs wait states the assembler does not add inside asm
            //  blocks; without them lanes 32.. received the PREVIOUS ballot)
            asm("s_nop 4\n\tv_writelane_b32 %0, %1, %3\n\tv_writelane_b32 %0, %2, %4" : "+v"(w) : "s"(lo), "s"(hi), "n"(j), "n"(32 + j));
        }
        OD_STAMP(6);
        if (slot < outCap) {
            if (lane < 8) reinterpret_cast<uint32_t *>(descOut + frame * descStride + (long long)slot * 32)[lane] = w;
            if (lane == 0) {
                RumiKeyPoint kp;
                kp.x = (float)x; kp.y = (float)y;
                const float2 lf = sLvF[level];
                if (level != 0) { kp.x = kp.x * lf.x; kp.y = kp.y * lf.x; }   // keypoint->pt *= scale (:1073-1075)
                kp.size = lf.y;
                kp.angle = angle;
                kp.response = (float)score;
                kp.octave = level;
                kp.class_id = -1;
                reinterpret_cast<RumiKeyPoint *>(reinterpret_cast<uint8_t *>(kpOut) + frame * kpStride)[slot] = kp;
            }
        }
    };

    const uint32_t *selP = selPacked + (long long)frame * selCap, *selM = selMeta + (long long)frame * selCap;
    auto key_rec = [&](int k, uint32_t &pk, uint32_t &mt) {     // key-point k of the frame: from k_assemble's arrays, or from this workgroup's own prologue
        if constexpr (kAssemble) { const int j = k - (kb - hw); pk = sOwnPk[j]; mt = sOwnMt[j]; }
        else { pk = selP[k]; mt = selM[k]; }
    };
    bool liveC = kb < cnt, liveN = kb + kKpPerWg < cnt;
    uint32_t pkC = 0, mtC = 0, pkN = 0, mtN = 0;
    if (liveC) key_rec(kb, pkC, mtC);
    if (liveN && kKpGroups > 1) key_rec(kb + kKpPerWg, pkN, mtN);
    if (kKpGroups == 1) liveN = false;
    Staged S;
    fetch_disc(pkC, mtC, liveC, S);
    fetch_patch(pkC, mtC, liveC, S);
#pragma unroll
    for (int g = 0; g < kKpGroups; g++) {
        if (!__any(liveC)) break;                                         // (key-points of a half wave come in ascending k: nothing further)
        OD_STAMP(0);
        stage_disc(liveC, S);
        OD_STAMP(1);
        // the key-point after this one: its pixels travel while this one is computed (the disc behind this one's disc store, the patch behind
        // this one's patch store: the registers are free then); the one after that: its record
        const int k2 = kb + (g + 2) * kKpPerWg;
        const bool liveNN = g + 2 < kKpGroups && k2 < cnt;
        uint32_t pkNN = 0, mtNN = 0;
        if (liveNN) key_rec(k2, pkNN, mtNN);
        if (g + 1 < kKpGroups) fetch_disc(pkN, mtN, liveN, S);
        OD_STAMP(2);
        float angle = 0.f;
        if (liveC) angle = orientation(pkC);
        stage_patch(liveC, S);                                            // (the same LDS rows: the moments above have read the disc)
        if (g + 1 < kKpGroups) fetch_patch(pkN, mtN, liveN, S);
        if (liveC) describe(pkC, mtC, angle);
        OD_STAMP(3);
        pkC = pkN; mtC = mtN; liveC = liveN;
        pkN = pkNN; mtN = mtNN; liveN = liveNN;
    }
#ifdef RUMI_OD_STAMP
    if (threadIdx.x == 0 && (blockIdx.x % 8) == 0 && blockIdx.y == 0) printf("od wg %d: loop-head %lld stage(wait loads) %lld issue-next %lld | IC_Angle %lld trig %lld rBRIEF %lld store %lld\n", (int)blockIdx.x, st[0], st[1], st[2], st[4], st[5], st[6], st[3]);
#endif
}

// ---- launch wrappers (called from orb_host.hip) ----
void launch_resize(const DevParams *dP, const DevParams &hP, ImgSrc src, const int16_t *coef, const RowTap *rowTab, int level, int nframes,
                   hipStream_t st, int32_t *clearWord) {
    static const int envRows = std::getenv("RUMI_RESIZE_ROWS") ? std::atoi(std::getenv("RUMI_RESIZE_ROWS")) : 0;
    const int rows = envRows ? envRows : (hP.lv[level].h >= 200 ? 8 : 4);
    dim3 g((hP.lv[level].w + 255) / 256, (hP.lv[level].h + 4 * rows - 1) / (4 * rows), nframes);
    if (rows == 8) hipLaunchKernelGGL(k_resize<8>, g, dim3(256), 0, st, dP, src, coef, rowTab, level, clearWord);
    else hipLaunchKernelGGL(k_resize<4>, g, dim3(256), 0, st, dP, src, coef, rowTab, level, clearWord);
}
void launch_pyramid_tiles(const DevParams *dP, ImgSrc src, const int16_t *coef, const RowTap *rowTab, const PyrTile *tiles, int ntiles, int bufBytes,
                          int tabEntries, int nframes, hipStream_t st, int32_t *clearWord, bool copyL0) {
    hipLaunchKernelGGL(k_pyramid_tiles, dim3(ntiles, nframes), dim3(256), (size_t)2 * bufBytes + (size_t)tabEntries * 8, st, dP, src, coef, rowTab, tiles, bufBytes, clearWord, copyL0 ? 1 : 0);
}
static FastLds fast_lds_of(const DevParams &hP) {
    // LDS per wave from the largest cell of this geometry
    int wMax = 0, hMax = 0;
    for (int l = 0; l < hP.nlevels; l++) { wMax = std::max(wMax, hP.lv[l].wCell); hMax = std::max(hMax, hP.lv[l].hCell); }
    FastLds F;
    F.tp = 4 * (((wMax + 3) >> 2) + 2);                   // the detection region's 4-pixel groups + one dword of margin on either side (tile column 4 = first detection column)
    F.sp = F.tp;                                          // the score map shares the tile's pitch (a pixel's score byte sits at its tile offset + a constant)
    F.tileBytes = (hMax + 6) * F.tp;
    F.scBytes = ((hMax + 2) * F.sp + 15) & ~15;
    F.maxIters = (wMax * hMax + 63) / 64 + 1;
    F.tileBytes = (F.tileBytes + 15) & ~15;
    // tile | score map | ring of (pixel, polarity) entries that passed the quick test (linear, kRingCap x uint16; the NMS ballots reuse it) |
    // list of scored pixels (kScoredCap x uint16)
    F.perWave = (F.tileBytes + F.scBytes + std::max(kRingCap * 2, F.maxIters * 8) + kScoredCap * 2 + 15) & ~15;
    return F;
}
void launch_fast(const DevParams *dP, const DevParams &hP, ImgSrc src, uint32_t *cellBuf, int32_t *cellCnt, int nframes,
                 hipStream_t st) {
    const FastLds F = fast_lds_of(hP);
    // tile pitches of the common image sizes as compile-time constants (cells up to 36 / 40 / 44 / 48 pixels wide: 44 / 48 / 52 / 56);
    // anything else takes the run-time instantiation
    const int wpg = 4;                                    // cells (= waves) per workgroup
    const dim3 grid((hP.totalCells + wpg - 1) / wpg, nframes);
    const size_t lds = (size_t)wpg * F.perWave;
#define RUMI_FAST_CASE(T) if (F.tp == T) { hipLaunchKernelGGL((k_fast_cells<T>), grid, dim3(64 * wpg), lds, st, dP, src, F, cellBuf, cellCnt); return; }
    RUMI_FAST_CASE(48) RUMI_FAST_CASE(44) RUMI_FAST_CASE(52) RUMI_FAST_CASE(56)
#undef RUMI_FAST_CASE
    hipLaunchKernelGGL((k_fast_cells<0>), grid, dim3(64 * wpg), lds, st, dP, src, F, cellBuf, cellCnt);
}
// workgroups per frame (each repeats the cheap scan and copies its share of the outputs: the copy is a chain of dependent LDS reads per
// element, so one workgroup per frame is ~40 us of latency whatever the batch)
static int compactSlices(int nframes) {
    static const int env = std::getenv("RUMI_COMPACT_SLICES") ? std::atoi(std::getenv("RUMI_COMPACT_SLICES")) : 0;
    return env > 0 ? env : (nframes < 32 ? 32 : 8);
}
void launch_compact(const DevParams *dP, const DevParams &hP, const uint32_t *cellBuf, const int32_t *cellCnt,
                    uint32_t *cand, int32_t *levelStart, int32_t *errFlag, int nframes, hipStream_t st) {
    hipLaunchKernelGGL(k_compact, dim3(nframes, compactSlices(nframes)), dim3(kCompactThreads), (hP.totalCells + 1) * sizeof(int), st, dP, cellBuf, cellCnt,
                       cand, levelStart, errFlag);
}
// strip width of a level's waves: 256 pixels unless that would put the row's partial dword into lane 0 or 1 of a wave (its mirrored bytes
// then lie in the previous wave) or make a wave's lane 63 need a halo dword that reaches beyond the row edge; narrower waves leave their
// last lanes as pure halo providers
static int blur_strip_width(int w) {
    for (int bw : {256, 240, 224, 208}) {
        const int r = w % bw;
        const bool partialInFirstLanes = r >= 1 && r <= 8;
        const bool lane63Halo = bw == 256 && (r >= 253 || r <= 3);
        if (!partialInFirstLanes && !lane63Halo) return bw;
    }
    return 192;
}
static BlurGrid blur_grid_of(const DevParams &hP, int rows, int *total) {
    BlurGrid G{};
    int run = 0;
    for (int l = 0; l < hP.nlevels; l++) {
        G.bw[l] = blur_strip_width(hP.lv[l].w);
        G.gx[l] = (hP.lv[l].w + G.bw[l] - 1) / G.bw[l];
        G.base[l] = run;
        run += G.gx[l] * ((hP.lv[l].h + 4 * rows - 1) / (4 * rows));
    }
    G.base[hP.nlevels] = run;
    *total = run;
    return G;
}
bool fast_blur_fusable(const DevParams &hP) { return fast_lds_of(hP).tp == 48; }
// FAST + blur of a few frames as one launch (k_fast_blur); false: this geometry has no fused instantiation, launch them separately
bool launch_fast_blur(const DevParams *dP, const DevParams &hP, ImgSrc src, uint32_t *cellBuf, int32_t *cellCnt, int nframes, int variant, hipStream_t st) {
    const FastLds F = fast_lds_of(hP);
    if (F.tp != 48) return false;                                  // (640 x 480 and its neighbours; other pitches keep the two launches)
    int run = 0;
    const BlurGrid G = blur_grid_of(hP, kBlurRowsSmall, &run);
    const int wpg = 4;
    const unsigned gxFast = (unsigned)((hP.totalCells + wpg - 1) / wpg);
    const dim3 grid(gxFast + (unsigned)run, nframes);
    const size_t lds = (size_t)wpg * F.perWave;
    if (variant) hipLaunchKernelGGL((k_fast_blur<48, 1>), grid, dim3(256), lds, st, dP, src, F, cellBuf, cellCnt, G, gxFast);
    else hipLaunchKernelGGL((k_fast_blur<48, 0>), grid, dim3(256), lds, st, dP, src, F, cellBuf, cellCnt, G, gxFast);
    return true;
}
void launch_blur(const DevParams *dP, const DevParams &hP, ImgSrc src, int nframes, int variant, hipStream_t st) {
    const bool small = nframes < 16;
    const int rows = small ? kBlurRowsSmall : kBlurRowsBatch;
    int run = 0;
    const BlurGrid G = blur_grid_of(hP, rows, &run);
    if (small) {
        if (variant) hipLaunchKernelGGL((k_blur<1, kBlurRowsSmall>), dim3(run, nframes), dim3(256), 0, st, dP, src, G);
        else hipLaunchKernelGGL((k_blur<0, kBlurRowsSmall>), dim3(run, nframes), dim3(256), 0, st, dP, src, G);
    } else {
        if (variant) hipLaunchKernelGGL((k_blur<1, kBlurRowsBatch>), dim3(run, nframes), dim3(256), 0, st, dP, src, G);
        else hipLaunchKernelGGL((k_blur<0, kBlurRowsBatch>), dim3(run, nframes), dim3(256), 0, st, dP, src, G);
    }
}
void launch_orient_desc(const DevParams *dP, ImgSrc src, const uint32_t *selPacked, const uint32_t *selMeta,
                        const int32_t *selCount, int selCap, int maxSel, RumiKeyPoint *kpOut, long long kpStride, uint8_t *descOut,
                        long long descStride, int outCap, int nframes, hipStream_t st) {
    if (maxSel <= 0) return;
    const int wg1 = (maxSel + kKpPerWg - 1) / kKpPerWg;
    const AssembleArgs none{};
    if ((long long)wg1 * nframes <= 2048)
        hipLaunchKernelGGL((k_orient_desc<1, false>), dim3(wg1, nframes), dim3(256), 0, st, dP, src, selPacked, selMeta, selCount, selCap, kpOut, kpStride, descOut,
                           descStride, outCap, none);
    else
        hipLaunchKernelGGL((k_orient_desc<2, false>), dim3((wg1 + 1) / 2, nframes), dim3(256), 0, st, dP, src, selPacked, selMeta, selCount, selCap, kpOut, kpStride,
                           descOut, descStride, outCap, none);
}
// k_assemble + k_orient_desc in ONE launch, for calls of a few frames (the caller guarantees (maxSel / 8) * nframes <= 2048 workgroups)
void launch_assemble_orient_desc(const DevParams *dP, ImgSrc src, const uint32_t *selLevel, const int32_t *selLevelCnt, int selLevelCap, int lap0, int lap1,
                                 int32_t *counts, long long countsStride, int32_t *errFlag, int32_t *errMirror, uint32_t *selPacked, uint32_t *selMeta,
                                 int32_t *selCount, int selCap, int maxSel, RumiKeyPoint *kpOut,
                                 long long kpStride, uint8_t *descOut, long long descStride, int outCap, int nframes, hipStream_t st) {
    if (maxSel <= 0) return;
    const int wg1 = (maxSel + kKpPerWg - 1) / kKpPerWg;
    const AssembleArgs A{selLevel, selLevelCnt, selLevelCap, lap0, lap1, counts, countsStride, errFlag, nframes == 1 ? errMirror : nullptr, selPacked, selMeta, selCount};
    hipLaunchKernelGGL((k_orient_desc<1, true>), dim3(wg1, nframes), dim3(256), 0, st, dP, src, (const uint32_t *)nullptr, (const uint32_t *)nullptr,
                       (const int32_t *)nullptr, selCap, kpOut, kpStride, descOut, descStride, outCap, A);
}

}  // namespace rumi
