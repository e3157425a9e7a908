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

__device__ __forceinline__ const uint8_t *level_base(const ImgSrc &s, const DevParams *P, int level, int frame,
                                                      int *pitch) {
    if (level == 0) {
        *pitch = s.l0Pitch;
        return s.l0 + (long long)frame * s.l0FrameStride;
    }
    *pitch = P->lv[level].pitch;
    return s.pyr + (long long)frame * P->arenaStride + P->lv[level].off;
}

// ------------------------------------------------------------------------------------------------
// Pyramid: one thread per destination pixel of level `level`; taps come from host-built tables that
// follow cv::resize's coefficient rule exactly (orb_geom.h).  HBM-bound: 1 B written, ~1.44 B read.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resize(const DevParams *__restrict__ P, ImgSrc src,
                                                const int16_t *__restrict__ coef, int level) {
    const DevLevel &D = P->lv[level];
    const DevLevel &S = P->lv[level - 1];
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int frame = blockIdx.z;
    if (dx >= D.w || dy >= D.h) return;
    int sp;
    const uint8_t *sb = level_base(src, P, level - 1, frame, &sp);
    uint8_t *db = src.pyr + (long long)frame * P->arenaStride + D.off;
    const int16_t *xofs = coef + D.coefX, *xa = xofs + D.w;
    const int16_t *yofs = coef + D.coefY, *ya = yofs + D.h;
    const int sx = xofs[dx], a0 = xa[dx * 2], a1 = xa[dx * 2 + 1];
    const int sy = yofs[dy], b0 = ya[dy * 2], b1 = ya[dy * 2 + 1];
    const int sy0 = sy >= 0 ? (sy < S.h ? sy : S.h - 1) : 0;
    const int sy1r = sy + 1;
    const int sy1 = sy1r >= 0 ? (sy1r < S.h ? sy1r : S.h - 1) : 0;
    const uint8_t *r0p = sb + (long long)sy0 * sp + sx, *r1p = sb + (long long)sy1 * sp + sx;
    int r0, r1;
    if (dx < D.xmax) {
        r0 = r0p[0] * a0 + r0p[1] * a1;
        r1 = r1p[0] * a0 + r1p[1] * a1;
    } else {
        r0 = r0p[0] * 2048;
        r1 = r1p[0] * 2048;
    }
    db[(long long)dy * D.pitch + dx] = (uint8_t)((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2);
}

// ------------------------------------------------------------------------------------------------
// FAST 9/16 + per-cell NMS + threshold fallback, one 256-thread workgroup per (frame, cell).
//
// score(p) = max over the 16 arcs of 9 of min |v - I_k| on the bright or dark side, minus 1  (cv's
// cornerScore with the start threshold folded out); p is a corner at T  <=>  score(p) >= T, so ONE
// score tile serves both thresholds.  NMS neighbours outside the cell's detection region count as 0,
// exactly as cv::FAST's zero-initialised score rows make them.
// The sub-image (<= 96x96 B) is staged in LDS; scores never touch HBM.
// ------------------------------------------------------------------------------------------------
constexpr int kTP = kCellTileMax + 4;                 // LDS pitch of the pixel tile
constexpr int kSP = kCellTileMax - 6 + 2;             // LDS pitch of the score tile (1-px zero ring)
constexpr int kMaxIters = (kCellTileMax - 6) * (kCellTileMax - 6) / 256 + 1;

__device__ __forceinline__ int min3i(int a, int b, int c) { return min(min(a, b), c); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }

__device__ __forceinline__ int fast_score_lds(const uint8_t *t /* centre, pitch kTP */) {
    const int v = t[0];
    int d[16];
    d[0] = v - t[3 * kTP];       d[1] = v - t[3 * kTP + 1];   d[2] = v - t[2 * kTP + 2];   d[3] = v - t[kTP + 3];
    d[4] = v - t[3];             d[5] = v - t[-kTP + 3];      d[6] = v - t[-2 * kTP + 2];  d[7] = v - t[-3 * kTP + 1];
    d[8] = v - t[-3 * kTP];      d[9] = v - t[-3 * kTP - 1];  d[10] = v - t[-2 * kTP - 2]; d[11] = v - t[-kTP - 3];
    d[12] = v - t[-3];           d[13] = v - t[kTP - 3];      d[14] = v - t[2 * kTP - 2];  d[15] = v - t[3 * kTP - 1];
    // windows of 3, then of 9 = three windows of 3 (indices mod 16)
    int lo3[16], hi3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        lo3[k] = min3i(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
        hi3[k] = max3i(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
    }
    int A = -256, Bn = 256;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        A = max(A, min3i(lo3[k], lo3[(k + 3) & 15], lo3[(k + 6) & 15]));
        Bn = min(Bn, max3i(hi3[k], hi3[(k + 3) & 15], hi3[(k + 6) & 15]));
    }
    return max(A, -Bn) - 1;
}

__global__ __launch_bounds__(256) void k_fast_cells(const DevParams *__restrict__ P, ImgSrc src,
                                                    uint32_t *__restrict__ cellBuf, int32_t *__restrict__ cellCnt) {
    __shared__ uint8_t tile[kCellTileMax * kTP];
    __shared__ uint8_t sc[(kCellTileMax - 6 + 2) * kSP];
    __shared__ unsigned long long balIni[kMaxIters][4], balMin[kMaxIters][4];
    __shared__ int prefix[kMaxIters][4];
    __shared__ int sTotal[2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cell = blockIdx.x, frame = blockIdx.y;
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
        if (tid == 0) cellCnt[cellIdx] = 0;
        return;
    }
    int pitch;
    const uint8_t *img = level_base(src, P, level, frame, &pitch) + (long long)iniY * pitch + iniX;
    for (int idx = tid; idx < rows * cols; idx += 256) {
        const int r = idx / cols, c = idx - r * cols;
        tile[r * kTP + c] = img[(long long)r * pitch + c];
    }
    const int dw = cols - 6, dh = rows - 6;
    for (int idx = tid; idx < (dw + 2) * (dh + 2); idx += 256) {
        const int r = idx / (dw + 2), c = idx - r * (dw + 2);
        sc[r * kSP + c] = 0;
    }
    __syncthreads();
    const int tlow = max(1, min(P->iniTh, P->minTh));
    const int npx = dw * dh;
    for (int idx = tid; idx < npx; idx += 256) {
        const int py = idx / dw, px = idx - py * dw;
        const int s = fast_score_lds(&tile[(py + 3) * kTP + px + 3]);
        if (s >= tlow) sc[(py + 1) * kSP + px + 1] = (uint8_t)s;
    }
    __syncthreads();
    const int iters = (npx + 255) >> 8;
    for (int it = 0; it < iters; it++) {
        const int idx = it * 256 + tid;
        bool isMax = false;
        int v = 0;
        if (idx < npx) {
            const int py = idx / dw, px = idx - py * dw;
            const uint8_t *s = &sc[(py + 1) * kSP + px + 1];
            v = s[0];
            isMax = v > 0 && v > s[-1] && v > s[1] && v > s[-kSP - 1] && v > s[-kSP] && v > s[-kSP + 1] &&
                    v > s[kSP - 1] && v > s[kSP] && v > s[kSP + 1];
        }
        const unsigned long long bi = __ballot(isMax && v >= P->iniTh);
        const unsigned long long bm = __ballot(isMax && v >= P->minTh);
        if (lane == 0) { balIni[it][wave] = bi; balMin[it][wave] = bm; }
    }
    __syncthreads();
    if (tid == 0) {
        int ti = 0;
        for (int it = 0; it < iters; it++)
            for (int w = 0; w < 4; w++) ti += __popcll(balIni[it][w]);
        const bool useMin = ti == 0;          // retry with minThFAST only if the first call found nothing
        int run = 0;
        for (int it = 0; it < iters; it++)
            for (int w = 0; w < 4; w++) {
                prefix[it][w] = run;
                run += __popcll(useMin ? balMin[it][w] : balIni[it][w]);
            }
        sTotal[0] = run;
        sTotal[1] = useMin;
        cellCnt[cellIdx] = run;
    }
    __syncthreads();
    const bool useMin = sTotal[1] != 0;
    uint32_t *out = cellBuf + cellIdx * P->maxCellCand;
    for (int it = 0; it < iters; it++) {
        const unsigned long long b = useMin ? balMin[it][wave] : balIni[it][wave];
        if ((b >> lane) & 1ull) {
            const int idx = it * 256 + tid;
            const int py = idx / dw, px = idx - py * dw;
            const int pos = prefix[it][wave] + __popcll(b & ((1ull << lane) - 1ull));
            const uint32_t x = (uint32_t)(px + 3 + ci_j * L.wCell), y = (uint32_t)(py + 3 + ci_i * L.hCell);
            out[pos] = x | (y << 12) | ((uint32_t)sc[(py + 1) * kSP + px + 1] << 24);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Candidate compaction: one workgroup per frame concatenates the cell lists in cell order (levels
// ascending, cells row-major) into cand[frame][...] and writes levelStart[frame][0..nlevels].
// A level that would exceed its capacity is truncated and flagged (levelStart keeps the true count in
// overflow[frame]); the host turns that into RUMI_E_CAPACITY.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_compact(const DevParams *__restrict__ P, const uint32_t *__restrict__ cellBuf,
                                                 const int32_t *__restrict__ cellCnt, uint32_t *__restrict__ cand,
                                                 int32_t *__restrict__ levelStart, int32_t *__restrict__ overflow) {
    extern __shared__ int sStart[];          // totalCells + 1 exclusive prefix
    __shared__ int part[256];
    const int tid = threadIdx.x, frame = blockIdx.x;
    const int nc = P->totalCells;
    const int32_t *cnt = cellCnt + (long long)frame * nc;
    const int chunk = (nc + 255) / 256;
    int sum = 0;
    for (int k = 0; k < chunk; k++) {
        const int c = tid * chunk + k;
        if (c < nc) sum += cnt[c];
    }
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < 256; i++) { const int t = part[i]; part[i] = run; run += t; }
        sStart[nc] = run;
    }
    __syncthreads();
    int run = part[tid];
    for (int k = 0; k < chunk; k++) {
        const int c = tid * chunk + k;
        if (c < nc) { sStart[c] = run; run += cnt[c]; }
    }
    __syncthreads();
    int32_t *ls = levelStart + (long long)frame * (kMaxLevels + 1);
    if (tid <= P->nlevels) {
        const int c = tid < P->nlevels ? P->lv[tid].cellBase : nc;
        ls[tid] = sStart[c];
    }
    if (tid < P->nlevels) {
        const int c0 = P->lv[tid].cellBase, c1 = c0 + P->lv[tid].nCells;
        if (sStart[c1] - sStart[c0] > P->lv[tid].candCap) atomicOr(&overflow[frame], 1 << tid);
    }
    uint32_t *out = cand + (long long)frame * P->totalCand;
    const int lane = tid & 63, wave = tid >> 6;
    for (int c = wave; c < nc; c += 4) {
        const int n = cnt[c], s0 = sStart[c];
        const uint32_t *in = cellBuf + ((long long)frame * nc + c) * P->maxCellCand;
        for (int k = lane; k < n; k += 64)
            if (s0 + k < P->totalCand) out[s0 + k] = in[k];
    }
}

// ------------------------------------------------------------------------------------------------
// Gaussian blur 7x7, sigma 2, fixed point: taps {18,34,48,56,48,34,18}/256, row pass to u16, column
// pass to u32, (v + 32768) >> 16.  One workgroup = 64 x 16 output pixels of one level of one frame;
// the 70 x 22 source patch (REFLECT_101 at the level's own edges) is staged in LDS.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

__global__ __launch_bounds__(256) void k_blur(const DevParams *__restrict__ P, ImgSrc src, int level) {
    __shared__ uint8_t in[22][72];
    __shared__ uint16_t hz[22][64];
    const DevLevel &L = P->lv[level];
    const int tid = threadIdx.x, frame = blockIdx.z;
    const int ox = blockIdx.x * 64, oy = blockIdx.y * 16;
    int pitch;
    const uint8_t *img = level_base(src, P, level, frame, &pitch);
    for (int idx = tid; idx < 22 * 70; idx += 256) {
        const int r = idx / 70, c = idx - r * 70;
        const int y = reflect101(oy + r - 3, L.h), x = reflect101(ox + c - 3, L.w);
        in[r][c] = img[(long long)y * pitch + x];
    }
    __syncthreads();
    for (int idx = tid; idx < 22 * 64; idx += 256) {
        const int r = idx >> 6, c = idx & 63;
        const uint8_t *p = &in[r][c];
        hz[r][c] = (uint16_t)(18 * (p[0] + p[6]) + 34 * (p[1] + p[5]) + 48 * (p[2] + p[4]) + 56 * p[3]);
    }
    __syncthreads();
    uint8_t *out = src.blur + (long long)frame * P->arenaStride + L.off;
    const int c = tid & 63;
    for (int r = tid >> 6; r < 16; r += 4) {
        const int x = ox + c, y = oy + r;
        if (x < L.w && y < L.h) {
            const uint32_t acc = 18u * (hz[r][c] + hz[r + 6][c]) + 34u * (hz[r + 1][c] + hz[r + 5][c]) +
                                 48u * (hz[r + 2][c] + hz[r + 4][c]) + 56u * hz[r + 3][c];
            out[(long long)y * L.pitch + x] = (uint8_t)((acc + 32768u) >> 16);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Orientation + descriptor + output assembly: one wave per selected key-point.
//   IC_Angle: integer moments over the radius-15 disc of the UN-blurred level, two disc rows per step;
//   rBRIEF:   lane l evaluates test pairs l, l+64, l+128, l+192; __ballot packs 64 bits at a time, which
//             is exactly the descriptor's little-endian bit order (bit k of byte i = pair 8i+k).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(256) void k_orient_desc(const DevParams *__restrict__ P, ImgSrc src,
                                                     const uint32_t *__restrict__ selPacked,
                                                     const uint32_t *__restrict__ selMeta,
                                                     const int32_t *__restrict__ selCount, int selCap,
                                                     RumiKeyPoint *__restrict__ kpOut, uint8_t *__restrict__ descOut,
                                                     int outCap) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6), frame = blockIdx.y;
    if (k >= selCount[frame]) return;
    const uint32_t pk = selPacked[(long long)frame * selCap + k], meta = selMeta[(long long)frame * selCap + k];
    const int level = meta & 0xFF, slot = (int)(meta >> 8);
    const DevLevel &L = P->lv[level];
    const int x = (int)(pk & 0xFFF) + kBorder, y = (int)((pk >> 12) & 0xFFF) + kBorder, score = (int)(pk >> 24);

    // IC_Angle (ORBextractor.cc:73-97)
    int pitch;
    const uint8_t *c = level_base(src, P, level, frame, &pitch) + (long long)y * pitch + x;
    const int half = lane >> 5, u = (lane & 31) - kHalfPatch;
    int m10 = 0, m01 = 0;
#pragma unroll 4
    for (int i = 0; i < 16; i++) {
        const int v = -kHalfPatch + 2 * i + half;
        const int av = v < 0 ? -v : v;
        if (v <= kHalfPatch && (lane & 31) < 31) {
            const int d = P->umax[av];
            if (u >= -d && u <= d) {
                const int val = c[(long long)v * pitch + u];
                m10 += u * val;
                m01 += v * val;
            }
        }
    }
    m10 = wave_sum(m10);
    m01 = wave_sum(m01);
    const float angle = fast_atan2_deg((float)m01, (float)m10);

    // computeOrbDescriptor (ORBextractor.cc:99-143) on the blurred level
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    const float ang = angle * factorPI;
    const float a = cosf_glibc(ang), b = sinf_glibc(ang);
    const uint8_t *bc = src.blur + (long long)frame * P->arenaStride + L.off + (long long)y * L.pitch + x;
    const int bp = L.pitch;
    unsigned long long bits[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int8_t *pt = &c_pattern[(j * 64 + lane) * 4];
        const float x0 = (float)pt[0], y0 = (float)pt[1], x1 = (float)pt[2], y1 = (float)pt[3];
        const int r0 = cv_round_f(x0 * b + y0 * a), c0 = cv_round_f(x0 * a - y0 * b);
        const int r1 = cv_round_f(x1 * b + y1 * a), c1 = cv_round_f(x1 * a - y1 * b);
        const int t0 = bc[r0 * bp + c0], t1 = bc[r1 * bp + c1];
        bits[j] = __ballot(t0 < t1);
    }
    if (slot < outCap) {
        if (lane < 4) {
            unsigned long long w = lane == 0 ? bits[0] : lane == 1 ? bits[1] : lane == 2 ? bits[2] : bits[3];
            reinterpret_cast<unsigned long long *>(descOut + ((long long)frame * outCap + slot) * 32)[lane] = w;
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
void launch_resize(const DevParams *dP, const DevParams &hP, ImgSrc src, const int16_t *coef, int level, int nframes,
                   hipStream_t st) {
    dim3 g((hP.lv[level].w + 63) / 64, (hP.lv[level].h + 3) / 4, nframes);
    hipLaunchKernelGGL(k_resize, g, dim3(256), 0, st, dP, src, coef, level);
}
void launch_fast(const DevParams *dP, const DevParams &hP, ImgSrc src, uint32_t *cellBuf, int32_t *cellCnt, int nframes,
                 hipStream_t st) {
    hipLaunchKernelGGL(k_fast_cells, dim3(hP.totalCells, nframes), dim3(256), 0, st, dP, src, cellBuf, cellCnt);
}
void launch_compact(const DevParams *dP, const DevParams &hP, const uint32_t *cellBuf, const int32_t *cellCnt,
                    uint32_t *cand, int32_t *levelStart, int32_t *overflow, int nframes, hipStream_t st) {
    hipLaunchKernelGGL(k_compact, dim3(nframes), dim3(256), (hP.totalCells + 1) * sizeof(int), st, dP, cellBuf, cellCnt,
                       cand, levelStart, overflow);
}
void launch_blur(const DevParams *dP, const DevParams &hP, ImgSrc src, int level, int nframes, hipStream_t st) {
    dim3 g((hP.lv[level].w + 63) / 64, (hP.lv[level].h + 15) / 16, nframes);
    hipLaunchKernelGGL(k_blur, g, dim3(256), 0, st, dP, src, level);
}
void launch_orient_desc(const DevParams *dP, ImgSrc src, const uint32_t *selPacked, const uint32_t *selMeta,
                        const int32_t *selCount, int selCap, int maxSel, RumiKeyPoint *kpOut, uint8_t *descOut,
                        int outCap, int nframes, hipStream_t st) {
    if (maxSel <= 0) return;
    hipLaunchKernelGGL(k_orient_desc, dim3((maxSel + 3) / 4, nframes), dim3(256), 0, st, dP, src, selPacked, selMeta,
                       selCount, selCap, kpOut, descOut, outCap);
}

}  // namespace rumi
