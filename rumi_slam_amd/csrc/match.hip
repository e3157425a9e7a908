// MI355X-native Hamming matchers behind include/rumi_match.h (kernels + host side).
//
// The reference walks map points / key-frame features one after another and lets each one see the
// assignments of the ones before it (ORBmatcher.cc:80-82, :248-249, :1556-1558).  GPU formulation, exact:
//   1. k_grid        Frame::AssignFeaturesToGrid as a key sort: (cell << 16 | feature) ascending, cell = ix*48+iy, so
//                    the cells GetFeaturesInArea visits for one ix are one contiguous range, already in its order.
//   2. k_queries_*   one query per map point / last-frame feature / key-frame feature (projection, window, levels).
//   3. k_candidates  one wave per query: enumerate candidates in the reference's order, 256-bit Hamming by
//                    xor + popcount, ballot-compacted into a per-query list (count pass, scan, fill pass).
//   4. k_resolve     one workgroup: every query picks its best candidate given "feature f is taken by an earlier
//                    query" (blockedFrom[f] = smallest blocking query index); iterate to the fix point.  After
//                    round k queries 0..k-1 hold their sequential result, and a fix point is the sequential result
//                    (induction on the query index), so the outcome equals the reference's loop bit for bit.
//                    Then the rotation histogram / ComputeThreeMaxima filter and the result arrays.
//   k_bruteforce     all-pairs best / second-best, queries in registers, train descriptors broadcast from LDS.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rumi_internal.h"
#include "rumi_common.h"
#include "rumi_match.h"

namespace rumi {

constexpr int kGridCols = 64, kGridRows = 48, kGridCells = kGridCols * kGridRows;   // Frame.h:42-43
constexpr int kMaxSortN = 16384;     // features per frame: the mono-initialisation extractor asks for 5 x nfeatures (Tracking.cc:581: 10 000 with TUM3.yaml)

enum { MODE_MAPPOINTS = 0, MODE_FRAME = 1, MODE_BOW = 2, MODE_BOW_KF = 3, MODE_SIM3 = 4, MODE_RELOC = 5, MODE_INIT = 6, MODE_FUSE = 7 };

struct Query {           // 48 bytes
    float u, v, r;       // window centre / half-size (MODE_BOW: unused)
    int32_t minLevel, maxLevel;
    int32_t valid;
    int32_t descId;      // row of the query descriptor in qDesc
    int32_t mpId;        // map point id this query assigns
    int32_t blocks;      // Observations() > 0: an assignment hides the feature from later queries
    int32_t c0, c1;      // MODE_BOW: candidate range in the frame's FeatureVector indices
    float angle;         // key-point angle on the query side (rotation histogram)
};

struct FrameDev {
    int n;
    const RumiKeyPoint *keys;
    const uint8_t *desc;
    float minX, minY, maxX, maxY, wInv, hInv;
    const float *scale;          // mvScaleFactors
    const uint16_t *sortedIdx;   // features sorted by (cell, index)
    const int32_t *cellStart;    // [kGridCells + 1]
};

__device__ __forceinline__ int hamming256(const uint32_t q[8], const uint32_t *d) {
    int s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) s += __popc(q[k] ^ d[k]);
    return s;
}

// inclusive prefix sum over the lanes of a wave by DPP (row prefix, row_bcast:15, row_bcast:31); lane 63 holds the total
__device__ __forceinline__ int wave_scan_incl_i32(int v) {
#define RUMI_DPP_ADD(ctl, rows) v += __builtin_amdgcn_update_dpp(0, v, ctl, rows, 0xf, false)
    RUMI_DPP_ADD(0x111, 0xf); RUMI_DPP_ADD(0x112, 0xf); RUMI_DPP_ADD(0x114, 0xf); RUMI_DPP_ADD(0x118, 0xf);
    RUMI_DPP_ADD(0x142, 0xa); RUMI_DPP_ADD(0x143, 0xc);
#undef RUMI_DPP_ADD
    return v;
}

// ---- 1. grid -----------------------------------------------------------------------------------------------
// Frame::AssignFeaturesToGrid as a counting sort by cell (cell = column-major ix*48+iy, the order GetFeaturesInArea walks),
// ascending key-point index inside a cell (= push_back order).  One workgroup; the per-cell segments (a handful of entries) are
// put in index order by an insertion sort after an unordered atomic placement.
// (nDev: the feature count where the extractor left it, when the host has not read it yet; n is then its upper bound)
__global__ __launch_bounds__(1024) void k_grid(int n, const RumiKeyPoint *__restrict__ keys, float minX, float minY, float wInv,
                                               float hInv, uint16_t *__restrict__ sortedIdx, int32_t *__restrict__ cellStart, const int32_t *__restrict__ nDev) {
    if (nDev) n = min(n, *nDev);
    __shared__ int32_t sCnt[kGridCells + 1];
    __shared__ uint16_t sCell[kMaxSortN], sOut[kMaxSortN];
    __shared__ int32_t sWave[16];
    const int tid = threadIdx.x;
    for (int c = tid; c <= kGridCells; c += 1024) sCnt[c] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        // Frame::PosInGrid: round() of the float expression, dropped when outside the grid
        const int px = (int)__builtin_roundf((keys[i].x - minX) * wInv);
        const int py = (int)__builtin_roundf((keys[i].y - minY) * hInv);
        uint16_t cell = 0xFFFF;
        if (px >= 0 && px < kGridCols && py >= 0 && py < kGridRows) { cell = (uint16_t)(px * kGridRows + py); atomicAdd(&sCnt[cell], 1); }
        sCell[i] = cell;
    }
    __syncthreads();
    // exclusive scan of the 3072 counts: 3 cells per thread, a DPP scan inside each wave, the 16 wave totals through LDS (one barrier)
    constexpr int kPer = (kGridCells + 1023) / 1024;
    int loc[kPer], sum = 0;
#pragma unroll
    for (int k = 0; k < kPer; k++) { const int c = tid * kPer + k; loc[k] = c < kGridCells ? sCnt[c] : 0; sum += loc[k]; }
    const int incl = wave_scan_incl_i32(sum);
    if ((tid & 63) == 63) sWave[tid >> 6] = incl;
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) { const int t = sWave[w]; total += t; if (w < (tid >> 6)) before += t; }
    int run = before + incl - sum;
#pragma unroll
    for (int k = 0; k < kPer; k++) {
        const int c = tid * kPer + k;
        if (c < kGridCells) { cellStart[c] = run; sCnt[c] = run; run += loc[k]; }
    }
    if (tid == 1023) cellStart[kGridCells] = total;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        const uint16_t cell = sCell[i];
        if (cell != 0xFFFF) sOut[atomicAdd(&sCnt[cell], 1)] = (uint16_t)i;
    }
    __syncthreads();
    // sCnt[c] is now the END of cell c; its start is the end of cell c-1 (or 0)
#pragma unroll
    for (int k = 0; k < kPer; k++) {
        const int c = tid * kPer + k;
        if (c >= kGridCells) continue;
        const int e = sCnt[c], b0 = e - loc[k];
        for (int i = b0 + 1; i < e; i++) {
            const uint16_t v = sOut[i];
            int j = i - 1;
            while (j >= b0 && sOut[j] > v) { sOut[j + 1] = sOut[j]; j--; }
            sOut[j + 1] = v;
        }
    }
    __syncthreads();
    for (int i = tid; i < total; i += 1024) sortedIdx[i] = sOut[i];
}

// ---- 2. queries ----------------------------------------------------------------------------------------------
// SearchByProjection(F, map points): ORBmatcher.cc:44-71
__device__ __forceinline__ Query mappoint_query(int i, bool inView, float px, float py, int lvl, float viewCos, float depth, bool isBad, int obs,
                                                const float *scaleFactors, float th, int farPoints, float thFar) {
    Query o{};
    o.valid = inView && !(farPoints && depth > thFar) && !isBad;
    if (o.valid) {
        float r = (double)viewCos > 0.998 ? 2.5f : 4.0f;      // RadiusByViewingCos (float vs double literal)
        if ((double)th != 1.0) r *= th;
        o.u = px; o.v = py;
        o.r = r * scaleFactors[lvl];
        o.minLevel = lvl - 1; o.maxLevel = lvl;
    }
    o.descId = i; o.mpId = i; o.blocks = obs > 0;
    return o;
}
__global__ void k_queries_mappoints(int nmp, const uint8_t *trackInView, const float *projX, const float *projY,
                                    const int32_t *scaleLevel, const float *viewCos, const float *trackDepth,
                                    const uint8_t *isBad, const int32_t *mpObs, const float *scaleFactors, float th,
                                    int farPoints, float thFar, Query *q) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nmp) return;
    q[i] = mappoint_query(i, trackInView[i] != 0, projX[i], projY[i], scaleLevel[i], viewCos[i], trackDepth[i], isBad[i] != 0, mpObs[i], scaleFactors, th, farPoints, thFar);
}

// SearchByProjection(Cur, Last): ORBmatcher.cc:1516-1551 (mono: levels nLastOctave-1 .. nLastOctave+1)
__global__ void k_queries_frame(int nlast, const RumiKeyPoint *lastKeys, const int32_t *lastMp, const uint8_t *lastOutlier,
                                const float *mpPos, const int32_t *mpObs, const float *Tcw, const float *K,
                                const float *scaleFactors, float th, float minX, float minY, float maxX, float maxY,
                                Query *q) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nlast) return;
    Query o{};
    const int mp = lastMp[i];
    if (mp >= 0 && !lastOutlier[i]) {
        // Sophus::SE3f * p: p + w*uv + q.vec x uv, uv = 2 (q.vec x p); then + t   (so3.hpp:358-367)
        const float qx = Tcw[0], qy = Tcw[1], qz = Tcw[2], qw = Tcw[3];
        const float p0 = mpPos[mp * 3], p1 = mpPos[mp * 3 + 1], p2 = mpPos[mp * 3 + 2];
        float u0 = qy * p2 - qz * p1, u1 = qz * p0 - qx * p2, u2 = qx * p1 - qy * p0;
        u0 += u0; u1 += u1; u2 += u2;
        const float c0 = qy * u2 - qz * u1, c1 = qz * u0 - qx * u2, c2 = qx * u1 - qy * u0;
        const float xc = ((p0 + qw * u0) + c0) + Tcw[4], yc = ((p1 + qw * u1) + c1) + Tcw[5], zc = ((p2 + qw * u2) + c2) + Tcw[6];
        const float invzc = (float)(1.0 / (double)zc);
        if (!(invzc < 0)) {
            const float u = K[0] * xc / zc + K[2], v = K[1] * yc / zc + K[3];      // Pinhole::project
            if (!(u < minX || u > maxX) && !(v < minY || v > maxY)) {
                const int oct = lastKeys[i].octave;
                o.valid = 1; o.u = u; o.v = v; o.r = th * scaleFactors[oct];
                o.minLevel = oct - 1; o.maxLevel = oct + 1;
            }
        }
        o.descId = mp; o.mpId = mp; o.blocks = mpObs[mp] > 0;
    }
    o.angle = lastKeys[i].angle;
    q[i] = o;
}

// SearchByBoW: one query per entry of the key-frame's FeatureVector, in (node, entry) order (ORBmatcher.cc:217-232).  One THREAD per entry
// (it finds its node by bisection of the offsets, then the node's twin in the frame's vector by bisection of the ids): a vocabulary level with
// few nodes -- levelsup near L, small trees -- used to leave the work to a handful of threads walking a hundred entries each (84 us at 10 nodes).
__global__ void k_queries_bow(int nnKF, const uint32_t *kfNodes, const int32_t *kfOff, const uint32_t *kfIdx,
                              const int32_t *kfMp, const uint8_t *mpBad, const RumiKeyPoint *kfKeys, int nnF,
                              const uint32_t *fNodes, const int32_t *fOff, Query *q, const int32_t *nnFdev = nullptr) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (nnKF <= 0 || p >= kfOff[nnKF]) return;
    if (nnFdev) nnF = *nnFdev;                              // the frame's FeatureVector was built on the device (k_fv_build)
    int a = 0, ahi = nnKF;                                  // the node that holds entry p: last a with kfOff[a] <= p
    while (ahi - a > 1) { const int mid = (a + ahi) >> 1; if (kfOff[mid] <= p) a = mid; else ahi = mid; }
    // the merge-walk of the two ordered maps visits exactly the node ids present in both
    int lo = 0, hi = nnF;
    const uint32_t id = kfNodes[a];
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (fNodes[mid] < id) lo = mid + 1; else hi = mid; }
    const bool hit = lo < nnF && fNodes[lo] == id;
    Query o{};
    const int feat = (int)kfIdx[p];
    const int mp = kfMp[feat];
    o.valid = hit && mp >= 0 && !mpBad[mp];
    o.descId = feat; o.mpId = mp; o.blocks = 1;
    if (o.valid) { o.c0 = fOff[lo]; o.c1 = fOff[lo + 1]; }
    o.angle = kfKeys[feat].angle;
    q[p] = o;
}

// MapPoint::PredictScale (MapPoint.cc:538-570); log in double (oracle/match_oracle.cc explains the choice)
__device__ __forceinline__ int predict_scale(float maxDistance, float dist, float logScaleFactor, int nLevels) {
    const float ratio = maxDistance / dist;
    int nScale = (int)ceil(log((double)ratio) / (double)logScaleFactor);
    if (nScale < 0) nScale = 0;
    else if (nScale >= nLevels) nScale = nLevels - 1;
    return nScale;
}
__device__ __forceinline__ void se3f_mul(const float *T, const float *p, float *o) {   // Sophus::SE3f * p (so3.hpp:358-367)
    const float qx = T[0], qy = T[1], qz = T[2], qw = T[3];
    float u0 = qy * p[2] - qz * p[1], u1 = qz * p[0] - qx * p[2], u2 = qx * p[1] - qy * p[0];
    u0 += u0; u1 += u1; u2 += u2;
    const float c0 = qy * u2 - qz * u1, c1 = qz * u0 - qx * u2, c2 = qx * u1 - qy * u0;
    o[0] = ((p[0] + qw * u0) + c0) + T[4]; o[1] = ((p[1] + qw * u1) + c1) + T[5]; o[2] = ((p[2] + qw * u2) + c2) + T[6];
}

// SearchByProjection(KeyFrame*, Sim3f&, points, ...): ORBmatcher.cc:389-436 (variant 0) / :491-539 (variant 1)
__global__ void k_queries_sim3(int nmp, const uint8_t *skip, const float *mpPos, const float *mpNormal, const float *mpMinDist,
                               const float *mpMaxDist, const float *pose /*Tcw7, K4, Ow3*/, const float *scaleFactors, int nLevels,
                               float logScaleFactor, float th, int variant, int blocks, int checkReproj, float minX, float minY, float maxX, float maxY,
                               Query *q) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nmp) return;
    Query o{};
    o.descId = i; o.mpId = i; o.blocks = blocks; o.c0 = checkReproj;
    const float *Tcw = pose, *K = pose + 7, *Ow = pose + 11;
    if (!skip[i]) {
        const float *p3Dw = mpPos + (size_t)i * 3;
        float pc[3];
        se3f_mul(Tcw, p3Dw, pc);
        if (!(pc[2] < 0.0f)) {
            float u, v;
            if (variant == 0) { u = K[0] * pc[0] / pc[2] + K[2]; v = K[1] * pc[1] / pc[2] + K[3]; }
            else { const float invz = 1 / pc[2]; const float x = pc[0] * invz, y = pc[1] * invz; u = K[0] * x + K[2]; v = K[1] * y + K[3]; }
            if (u >= minX && u < maxX && v >= minY && v < maxY) {                       // KeyFrame::IsInImage
                const float maxD = 1.2f * mpMaxDist[i], minD = 0.8f * mpMinDist[i];
                const float P0 = p3Dw[0] - Ow[0], P1 = p3Dw[1] - Ow[1], P2 = p3Dw[2] - Ow[2];
                const float dist = sqrtf((P0 * P0 + P1 * P1) + P2 * P2);
                const float *Pn = mpNormal + (size_t)i * 3;
                if (!(dist < minD || dist > maxD) && !((double)((P0 * Pn[0] + P1 * Pn[1]) + P2 * Pn[2]) < 0.5 * (double)dist)) {
                    const int lvl = predict_scale(mpMaxDist[i], dist, logScaleFactor, nLevels);
                    o.valid = 1; o.u = u; o.v = v; o.r = th * scaleFactors[lvl];
                    o.minLevel = lvl - 1; o.maxLevel = lvl;                              // the level test of :445-448 / :553-556
                }
            }
        }
    }
    q[i] = o;
}

// SearchBySim3, one direction (ORBmatcher.cc:1329-1371 / :1405-1447): points already in the target camera frame
__global__ void k_queries_campoints(int n, const uint8_t *skip, const float *pc, const float *mpMinDist, const float *mpMaxDist, const float *K,
                                    const float *scaleFactors, int nLevels, float logScaleFactor, float th, float minX, float minY, float maxX,
                                    float maxY, Query *q) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Query o{};
    o.descId = i; o.mpId = i;
    if (!skip[i]) {
        const float *p = pc + (size_t)i * 3;
        if (!((double)p[2] < 0.0)) {
            const float invz = (float)(1.0 / (double)p[2]);
            const float x = p[0] * invz, y = p[1] * invz;
            const float u = K[0] * x + K[2], v = K[1] * y + K[3];
            if (u >= minX && u < maxX && v >= minY && v < maxY) {                       // KeyFrame::IsInImage
                const float maxD = 1.2f * mpMaxDist[i], minD = 0.8f * mpMinDist[i];
                const float dist = sqrtf((p[0] * p[0] + p[1] * p[1]) + p[2] * p[2]);
                if (!(dist < minD || dist > maxD)) {
                    const int lvl = predict_scale(mpMaxDist[i], dist, logScaleFactor, nLevels);
                    o.valid = 1; o.u = u; o.v = v; o.r = th * scaleFactors[lvl];
                    o.minLevel = lvl - 1; o.maxLevel = lvl;
                }
            }
        }
    }
    q[i] = o;
}

// SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist): ORBmatcher.cc:1700-1733
__global__ void k_queries_reloc(int nkf, const RumiKeyPoint *kfKeys, const int32_t *kfMp, const uint8_t *skip, const float *mpPos,
                                const float *mpMinDist, const float *mpMaxDist, const float *pose, const float *scaleFactors, int nLevels,
                                float logScaleFactor, float th, float minX, float minY, float maxX, float maxY, Query *q) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nkf) return;
    Query o{};
    const int mp = kfMp[i];
    const float *Tcw = pose, *K = pose + 7, *Ow = pose + 11;
    if (mp >= 0 && !skip[mp]) {
        const float *xw = mpPos + (size_t)mp * 3;
        float pc[3];
        se3f_mul(Tcw, xw, pc);
        const float u = K[0] * pc[0] / pc[2] + K[2], v = K[1] * pc[1] / pc[2] + K[3];
        if (!(u < minX || u > maxX) && !(v < minY || v > maxY)) {
            const float P0 = xw[0] - Ow[0], P1 = xw[1] - Ow[1], P2 = xw[2] - Ow[2];
            const float dist3D = sqrtf((P0 * P0 + P1 * P1) + P2 * P2);
            const float maxD = 1.2f * mpMaxDist[mp], minD = 0.8f * mpMinDist[mp];
            if (!(dist3D < minD || dist3D > maxD)) {
                const int lvl = predict_scale(mpMaxDist[mp], dist3D, logScaleFactor, nLevels);
                o.valid = 1; o.u = u; o.v = v; o.r = th * scaleFactors[lvl];
                o.minLevel = lvl - 1; o.maxLevel = lvl + 1;
            }
        }
        o.descId = mp; o.mpId = mp; o.blocks = 1;
    }
    o.angle = kfKeys[i].angle;
    q[i] = o;
}

// SearchForInitialization: level-0 key-points of F1, window around vbPrevMatched (ORBmatcher.cc:593-602)
__global__ void k_queries_init(int n1, const RumiKeyPoint *keys1, const float *prevMatched, float windowSize, Query *q) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n1) return;
    Query o{};
    o.valid = !(keys1[i].octave > 0);
    o.u = prevMatched[2 * i]; o.v = prevMatched[2 * i + 1]; o.r = windowSize;
    o.minLevel = 0; o.maxLevel = 0;
    o.descId = i; o.mpId = i; o.angle = keys1[i].angle;
    q[i] = o;
}

// Frame::isInFrustum (Frame.cc:558-617, mono): one lane per map point
__global__ void k_is_in_frustum(int nmp, const float *pose /*Rcw9 tcw3 Ow3 K4*/, float minX, float minY, float maxX, float maxY,
                                float logScaleFactor, int nLevels, float viewingCosLimit, const float *mpPos, const float *mpNormal,
                                const float *mpMinDist, const float *mpMaxDist, uint8_t *inView, float *projX, float *projY,
                                int32_t *scaleLevel, float *viewCosOut, float *trackDepth, const uint8_t *skip = nullptr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nmp) return;
    if (skip && skip[i]) {                                // SearchLocalPoints does not evaluate these (already matched in this frame / bad)
        inView[i] = 0; projX[i] = -1; projY[i] = -1; scaleLevel[i] = 0; viewCosOut[i] = 0; trackDepth[i] = 0;
        return;
    }
    const float *R = pose, *t = pose + 9, *Ow = pose + 12, *K = pose + 15;
    const float *P = mpPos + (size_t)i * 3;
    uint8_t in = 0;
    float px = -1, py = -1, vc = 0, depth = 0;
    int lvl = 0;
    float Pc[3];
#pragma unroll
    for (int r = 0; r < 3; r++) Pc[r] = ((R[r * 3] * P[0] + R[r * 3 + 1] * P[1]) + R[r * 3 + 2] * P[2]) + t[r];
    const float Pc_dist = sqrtf((Pc[0] * Pc[0] + Pc[1] * Pc[1]) + Pc[2] * Pc[2]);
    if (!(Pc[2] < 0.0f)) {
        const float u = K[0] * Pc[0] / Pc[2] + K[2], v = K[1] * Pc[1] / Pc[2] + K[3];
        if (!(u < minX || u > maxX) && !(v < minY || v > maxY)) {
            px = u; py = v;
            const float maxD = 1.2f * mpMaxDist[i], minD = 0.8f * mpMinDist[i];
            const float P0 = P[0] - Ow[0], P1 = P[1] - Ow[1], P2 = P[2] - Ow[2];
            const float dist = sqrtf((P0 * P0 + P1 * P1) + P2 * P2);
            if (!(dist < minD || dist > maxD)) {
                const float *Pn = mpNormal + (size_t)i * 3;
                const float viewCos = ((P0 * Pn[0] + P1 * Pn[1]) + P2 * Pn[2]) / dist;
                if (!(viewCos < viewingCosLimit)) {
                    lvl = predict_scale(mpMaxDist[i], dist, logScaleFactor, nLevels);
                    in = 1; depth = Pc_dist; vc = viewCos;
                }
            }
        }
    }
    inView[i] = in; projX[i] = px; projY[i] = py; scaleLevel[i] = lvl; viewCosOut[i] = vc; trackDepth[i] = depth;
}

// ---- uploads: one pinned block per call, scattered to the arrays on the device ---------------------------------------------
struct Segment { void *dst; uint32_t off, bytes; };
constexpr int kMaxSegments = 32;
__global__ __launch_bounds__(256) void k_scatter(const uint8_t *__restrict__ mirror, int nseg) {
    const Segment sg = reinterpret_cast<const Segment *>(mirror)[blockIdx.y];
    if ((int)blockIdx.y >= nseg) return;
    const uint32_t words = sg.bytes >> 2;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(mirror + sg.off);
    uint32_t *dst = reinterpret_cast<uint32_t *>(sg.dst);
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < words; i += gridDim.x * 256) dst[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x < (sg.bytes & 3)) {
        const uint32_t k = (words << 2) + threadIdx.x;
        reinterpret_cast<uint8_t *>(sg.dst)[k] = mirror[sg.off + k];
    }
}

// ---- 3. candidates: one wave per query -----------------------------------------------------------------------------
// list entry: feature (16 bit) | distance (9 bit) << 16 | octave (4 bit) << 25
// Lists of up to kSortMax entries are stored SORTED by (distance, position in the reference's candidate order): the
// reference's "best / second best among the candidates not yet taken" is then simply the first / second not-taken entry
// (strict `<` keeps the earliest of equal distances, and a displaced best becomes the second), so a resolve round reads a
// couple of entries per query instead of the whole list.  Longer lists stay in candidate order and are scanned in full.
constexpr int kSortMax = 1024;
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int PASS> __device__ __forceinline__ int offsets_of(const int32_t *offsets, int qi, int listCap) { return PASS == 2 ? qi * listCap : offsets[qi]; }

// The candidates of one query, walked by one wave: FILL = false counts them, FILL = true computes their Hamming distances and stores them (into the
// wave's LDS sort arrays when they fit, else straight to `out`).  Returns the count.
template <bool FILL>
__device__ __forceinline__ int candidates_walk(int mode, const Query &Q, const FrameDev &F, const uint32_t (&qd)[8], const uint32_t *__restrict__ fvIdx, bool sorted,
                                               uint32_t *out, uint32_t *key, uint32_t *val, int lane) {
    int count = 0;
    if (mode == MODE_BOW || mode == MODE_BOW_KF) {
        for (int p = Q.c0 + lane; p - lane < Q.c1; p += 64) {
            const bool ok = p < Q.c1;
            if (FILL && ok) {
                const int idx = (int)fvIdx[p];
                const int d = hamming256(qd, reinterpret_cast<const uint32_t *>(F.desc + (size_t)idx * 32));
                const uint32_t e = (uint32_t)idx | ((uint32_t)d << 16);
                if (sorted) { key[p - Q.c0] = ((uint32_t)d << 10) | (uint32_t)(p - Q.c0); val[p - Q.c0] = e; }
                else out[p - Q.c0] = e;
            }
        }
        return Q.c1 - Q.c0;
    }
    // Frame::GetFeaturesInArea (Frame.cc:695-750)
    const int nMinCellX = max(0, (int)floorf((Q.u - F.minX - Q.r) * F.wInv));
    const int nMaxCellX = min(kGridCols - 1, (int)ceilf((Q.u - F.minX + Q.r) * F.wInv));
    const int nMinCellY = max(0, (int)floorf((Q.v - F.minY - Q.r) * F.hInv));
    const int nMaxCellY = min(kGridRows - 1, (int)ceilf((Q.v - F.minY + Q.r) * F.hInv));
    if (nMinCellX < kGridCols && nMaxCellX >= 0 && nMinCellY < kGridRows && nMaxCellY >= 0) {
        const bool checkLevels = Q.minLevel > 0 || Q.maxLevel >= 0;
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
            const int p0 = F.cellStart[ix * kGridRows + nMinCellY], p1 = F.cellStart[ix * kGridRows + nMaxCellY + 1];
            for (int base = p0; base < p1; base += 64) {
                const int p = base + lane;
                bool pass = false;
                int idx = 0, oct = 0;
                if (p < p1) {
                    idx = F.sortedIdx[p];
                    const RumiKeyPoint kp = F.keys[idx];
                    oct = kp.octave;
                    pass = true;
                    if (checkLevels) {
                        if (oct < Q.minLevel) pass = false;
                        if (Q.maxLevel >= 0 && oct > Q.maxLevel) pass = false;
                    }
                    const float dx = kp.x - Q.u, dy = kp.y - Q.v;
                    if (!(fabsf(dx) < Q.r && fabsf(dy) < Q.r)) pass = false;
                    if (mode == MODE_FUSE && Q.c0 && pass) {                        // mono reprojection gate, ORBmatcher.cc:1138-1145
                        const float ex = Q.u - kp.x, ey = Q.v - kp.y;
                        const float e2 = ex * ex + ey * ey;
                        const float s2 = F.scale[oct] * F.scale[oct];              // mvLevelSigma2; mvInvLevelSigma2 = 1.0f / it
                        if ((double)(e2 * (1.0f / s2)) > 5.99) pass = false;
                    }
                }
                const unsigned long long b = __ballot(pass);
                if (FILL && pass) {
                    const int d = hamming256(qd, reinterpret_cast<const uint32_t *>(F.desc + (size_t)idx * 32));
                    const int pos = count + __popcll(b & ((1ull << lane) - 1ull));
                    const uint32_t e = (uint32_t)idx | ((uint32_t)d << 16) | ((uint32_t)(oct & 15) << 25);
                    if (sorted) { key[pos] = ((uint32_t)d << 10) | (uint32_t)pos; val[pos] = e; }
                    else out[pos] = e;
                }
                count += __popcll(b);
            }
        }
    }
    return count;
}

// bitonic sort of one wave's (key, val) pairs in LDS by key, then the values to `out`
__device__ __forceinline__ void wave_bitonic_store(uint32_t *key, uint32_t *val, int total, uint32_t *out, int lane) {
    int m = 1;
    while (m < total) m <<= 1;
    for (int i = total + lane; i < m; i += 64) key[i] = 0xFFFFFFFFu;
    wave_lds_fence();
    for (int k = 2; k <= m; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < (m >> 1); t += 64) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                const uint32_t a = key[i], b = key[l];
                if ((a > b) == ((i & k) == 0)) {
                    key[i] = b; key[l] = a;
                    const uint32_t va = val[i]; val[i] = val[l]; val[l] = va;
                }
            }
            wave_lds_fence();
        }
    for (int i = lane; i < total; i += 64) out[i] = val[i];
}

// PASS 0: count pass (counts[q]).  PASS 1: fill pass at the offsets a scan of the counts produced.  PASS 2: both in one launch, every query's list
// in a fixed slot of `listCap` entries (offsets[q] = q * listCap written here): two dispatches (~4.5 us each) less per search; a query with more
// candidates than a slot raises kFusedOverflow and the host repeats the search with passes 0 / scan / 1.
constexpr int kFusedOverflow = -0x40000000;
template <int PASS>
__global__ __launch_bounds__(256) void k_candidates(int mode, int nq, const Query *__restrict__ q, FrameDev F,
                                                    const uint8_t *__restrict__ qDesc, const uint32_t *__restrict__ fvIdx,
                                                    int32_t *__restrict__ counts, int32_t *__restrict__ offsets,
                                                    uint32_t *__restrict__ lists, int listCap, int32_t *__restrict__ overflow) {
    constexpr bool FILL = PASS != 0;
    __shared__ uint32_t sKey[FILL ? 4 * kSortMax : 1], sVal[FILL ? 4 * kSortMax : 1];
    const int lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (qi >= nq) return;
    if (PASS == 1 && offsets[nq] > listCap) {            // the arena cannot hold this call's lists: report the need, write nothing
        if (qi == 0 && lane == 0) *overflow = offsets[nq];
        return;
    }
    const Query Q = q[qi];
    if (PASS == 2 && lane == 0) offsets[qi] = qi * listCap;
    if (!Q.valid) {
        if (PASS != 1 && lane == 0) counts[qi] = 0;
        return;
    }
    uint32_t *key = sKey + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * kSortMax, *val = sVal + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * kSortMax;
    uint32_t qd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int total = 0;
    if (PASS == 2 && mode != MODE_BOW && mode != MODE_BOW_KF) {
        // The usual search of the Tracking thread (a window of a few grid columns, a handful of candidates) as ONE dependent chain
        // instead of two walks of four loads per column: the cell ranges of all columns at once (one lane each), the window's
        // feature slots flat over the lanes (slot -> column by the prefix of the range lengths: the reference's candidate order),
        // key-point and descriptor of every slot fetched together, and up to 64 candidates ranked in registers.
        const int nMinCellX = max(0, (int)floorf((Q.u - F.minX - Q.r) * F.wInv));
        const int nMaxCellX = min(kGridCols - 1, (int)ceilf((Q.u - F.minX + Q.r) * F.wInv));
        const int nMinCellY = max(0, (int)floorf((Q.v - F.minY - Q.r) * F.hInv));
        const int nMaxCellY = min(kGridRows - 1, (int)ceilf((Q.v - F.minY + Q.r) * F.hInv));
        int ncol = 0;
        if (nMinCellX < kGridCols && nMaxCellX >= 0 && nMinCellY < kGridRows && nMaxCellY >= 0 && nMaxCellY >= nMinCellY) ncol = max(0, nMaxCellX - nMinCellX + 1);
        int c0 = 0, len = 0;
        if (lane < ncol) {
            const int cell = (nMinCellX + lane) * kGridRows;
            c0 = F.cellStart[cell + nMinCellY];
            len = F.cellStart[cell + nMaxCellY + 1] - c0;
        }
        const uint32_t qmine = reinterpret_cast<const uint32_t *>(qDesc + (size_t)Q.descId * 32)[lane & 7];
        const int incl = wave_scan_incl_i32(len);
        const int T = __builtin_amdgcn_readlane(incl, 63);
        if (T <= kSortMax) {
#pragma unroll
            for (int k = 0; k < 8; k++) qd[k] = __shfl(qmine, k);
            const bool checkLevels = Q.minLevel > 0 || Q.maxLevel >= 0;
            uint32_t *out = lists + qi * listCap;
            int count = 0;
            uint32_t myKey = 0xFFFFFFFFu, myVal = 0;
            unsigned long long b = 0;
            for (int base = 0; base < T; base += 64) {
                const int sl = base + lane;
                const bool live = sl < T;
                int col = 0;
                for (int c = 0; c < ncol; c++) col += __builtin_amdgcn_readlane(incl, c) <= sl;
                const int cc = live ? col : 0;
                const int p = __shfl(c0, cc) + (sl - (__shfl(incl, cc) - __shfl(len, cc)));
                bool pass = false;
                int idx = 0, oct = 0, d = 0;
                if (live) {
                    idx = F.sortedIdx[p];
                    const RumiKeyPoint kp = F.keys[idx];
                    const uint4 *dp = reinterpret_cast<const uint4 *>(F.desc + (size_t)idx * 32);
                    const uint4 d0 = dp[0], d1 = dp[1];
                    oct = kp.octave;
                    pass = true;
                    if (checkLevels) {
                        if (oct < Q.minLevel) pass = false;
                        if (Q.maxLevel >= 0 && oct > Q.maxLevel) pass = false;
                    }
                    const float dx = kp.x - Q.u, dy = kp.y - Q.v;
                    if (!(fabsf(dx) < Q.r && fabsf(dy) < Q.r)) pass = false;
                    if (mode == MODE_FUSE && Q.c0 && pass) {                        // mono reprojection gate, ORBmatcher.cc:1138-1145
                        const float ex = Q.u - kp.x, ey = Q.v - kp.y;
                        const float e2 = ex * ex + ey * ey;
                        const float s2 = F.scale[oct] * F.scale[oct];              // mvLevelSigma2; mvInvLevelSigma2 = 1.0f / it
                        if ((double)(e2 * (1.0f / s2)) > 5.99) pass = false;
                    }
                    d = __popc(qd[0] ^ d0.x) + __popc(qd[1] ^ d0.y) + __popc(qd[2] ^ d0.z) + __popc(qd[3] ^ d0.w) +
                        __popc(qd[4] ^ d1.x) + __popc(qd[5] ^ d1.y) + __popc(qd[6] ^ d1.z) + __popc(qd[7] ^ d1.w);
                }
                b = __ballot(pass);
                if (pass) {
                    const int pos = count + __popcll(b & ((1ull << lane) - 1ull));
                    myKey = ((uint32_t)d << 10) | (uint32_t)pos;
                    myVal = (uint32_t)idx | ((uint32_t)d << 16) | ((uint32_t)(oct & 15) << 25);
                    if (T > 64) { key[pos] = myKey; val[pos] = myVal; }
                }
                count += __popcll(b);
            }
            if (lane == 0) counts[qi] = count;
            if (count > listCap) {
                if (lane == 0) atomicExch(overflow, kFusedOverflow);
                return;
            }
            if (T <= 64) {                                                  // one trip: rank among the passing lanes (keys are distinct)
                int rank = 0;
                for (unsigned long long bb = b; bb; bb &= bb - 1) {
                    const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)myKey, __builtin_ctzll(bb));
                    rank += kj < myKey;
                }
                if (myKey != 0xFFFFFFFFu) out[rank] = myVal;
            } else if (count > 0) {
                wave_bitonic_store(key, val, count, out, lane);
            }
            return;
        }
    }
    if (PASS == 0 || PASS == 2) {
        total = candidates_walk<false>(mode, Q, F, qd, fvIdx, false, nullptr, key, val, lane);
        if (lane == 0) counts[qi] = total;
        if (PASS == 0) return;
        if (total > listCap) {
            if (lane == 0) atomicExch(overflow, kFusedOverflow);
            return;
        }
    } else total = counts[qi];
    const bool sorted = total <= kSortMax;
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(qDesc + (size_t)Q.descId * 32);
        const uint32_t mine = src[lane & 7];
#pragma unroll
        for (int k = 0; k < 8; k++) qd[k] = __shfl(mine, k);
    }
    uint32_t *out = lists + offsets_of<PASS>(offsets, qi, listCap);
    candidates_walk<true>(mode, Q, F, qd, fvIdx, sorted, out, key, val, lane);
    if (sorted && total > 0) wave_bitonic_store(key, val, total, out, lane);
}

// exclusive scan of counts -> offsets (single workgroup; nq is a few thousand)
__global__ __launch_bounds__(256) void k_scan(int n, const int32_t *__restrict__ counts, int32_t *__restrict__ offsets) {
    __shared__ int part[256];
    const int tid = threadIdx.x, chunk = (n + 255) / 256;
    int s = 0;
    for (int k = 0; k < chunk; k++) { const int i = tid * chunk + k; if (i < n) s += counts[i]; }
    part[tid] = s;
    __syncthreads();
    if (tid == 0) { int run = 0; for (int i = 0; i < 256; i++) { const int t = part[i]; part[i] = run; run += t; } offsets[n] = run; }
    __syncthreads();
    int run = part[tid];
    for (int k = 0; k < chunk; k++) { const int i = tid * chunk + k; if (i < n) { offsets[i] = run; run += counts[i]; } }
}

// ---- 4. resolve ------------------------------------------------------------------------------------------------
struct ResolveArgs {
    int mode, nq, nfeat;
    const Query *q;
    const int32_t *counts, *offsets;
    const uint32_t *lists;
    const RumiKeyPoint *featKeys;   // angles of the frame's key-points (rotation histogram)
    const int32_t *mpObs;           // Observations() per map point id (initial occupancy), may be null (BOW)
    int32_t *featMp;                // in: initial frame_mp (MODE 0/1); out: final ids   [nfeat]
    int32_t *assign;                // scratch [nq]: feature chosen by each query or -1
    int32_t *nmatches;              // out
    float nnratio;
    int checkOri;
    const uint8_t *featBlocked0;   // optional [nfeat]: feature unavailable from the start (overrides the featMp/mpObs rule)
    float thrF;                    // MODE_SIM3: TH_LOW * ratioHamming
    int thrI;                      // MODE_RELOC: ORBdist
    const int32_t *overflow;       // set by the fill pass when the list arena is too small: nothing to resolve
    // optional tail (the Tracking step): PoseOptimization's correspondences gathered from the vector this search leaves (k_track_gather's work,
    // one launch less between the search and the optimisation); gXw == nullptr: none
    const float *gMpPos, *gInvSigma2;
    float *gXw, *gObs, *gW;
    int32_t *gIdx, *gStart, *gSnapshot;
};

__device__ __forceinline__ int rot_bin(float a, float b) {          // ORBmatcher.cc:1592-1599
    const float factor = 1.0f / RUMI_HISTO_LENGTH;
    float rot = a - b;
    if (rot < 0.0f) rot += 360.0f;
    int bin = (int)__builtin_roundf(rot * factor);
    if (bin == RUMI_HISTO_LENGTH) bin = 0;
    return bin;
}

// Correspondences of Optimizer::PoseOptimization(Frame*) (Optimizer.cc:749-815, mono): the features with a map point, in feature order.
// One workgroup of 1024 threads, ordered compaction (ballot + wave offsets through LDS, chunks of 1024 features).
__device__ __forceinline__ void gather_correspondences(int n, const RumiKeyPoint *__restrict__ keys, const int32_t *featMp, const float *__restrict__ mpPos,
                                                       const float *__restrict__ invSigma2, float *Xw, float *obs, float *w, int32_t *idx, int32_t *start,
                                                       int32_t *snapshot, int *sWave /* [16] */, int *sBase) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) *sBase = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += 1024) {
        const int i = c0 + tid;
        const int mp = i < n ? featMp[i] : -1;
        if (snapshot && i < n) snapshot[i] = mp;            // the frame's map-point vector as the search left it (the optimisation's outliers leave it next)
        const unsigned long long b = __ballot(mp >= 0);
        if (lane == 0) sWave[wave] = __popcll(b);
        __syncthreads();
        int off = *sBase;
        for (int k = 0; k < wave; k++) off += sWave[k];
        if (mp >= 0) {
            const int c = off + __popcll(b & ((1ull << lane) - 1));
            Xw[3 * c] = mpPos[3 * mp]; Xw[3 * c + 1] = mpPos[3 * mp + 1]; Xw[3 * c + 2] = mpPos[3 * mp + 2];
            obs[2 * c] = keys[i].x; obs[2 * c + 1] = keys[i].y;
            w[c] = invSigma2[keys[i].octave];
            idx[c] = i;
        }
        __syncthreads();
        if (tid == 0) { int t = *sBase; for (int k = 0; k < 16; k++) t += sWave[k]; *sBase = t; }
        __syncthreads();
    }
    if (tid == 0) { start[0] = 0; start[1] = *sBase; }
}

// wave-wide maximum / sum of one 32-bit value by DPP (row prefix, row_bcast:15, row_bcast:31; the total sits in lane 63)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#define RUMI_DPP_MAX(ctl, rows) v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctl, rows, 0xf, false))
    RUMI_DPP_MAX(0x111, 0xf); RUMI_DPP_MAX(0x112, 0xf); RUMI_DPP_MAX(0x114, 0xf); RUMI_DPP_MAX(0x118, 0xf);
    RUMI_DPP_MAX(0x142, 0xa); RUMI_DPP_MAX(0x143, 0xc);
#undef RUMI_DPP_MAX
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ int wave_sum_i32(int v) { return __builtin_amdgcn_readlane(wave_scan_incl_i32(v), 63); }

// One workgroup iterates "every query picks its best candidate among the features no EARLIER query holds" to its fixed point (the
// result of the reference's sequential loop).  A round is latency, not work: what a round needs of a query -- count, the head of its
// candidate list, the blocks flag, its current pick -- is read once into registers (the first kResQ queries of a thread, i.e. up to
// 2048 queries; the rest go through global memory as before), the initial occupancy of a thread's features is a bit mask, and a
// round is three barriers over LDS.
constexpr int kResQ = 2, kResK = 4;
__global__ __launch_bounds__(1024) void k_resolve(ResolveArgs A) {
    extern __shared__ int32_t blockedFrom[];      // [nfeat] smallest blocking query index; -1 = taken before the call
    __shared__ int sChanged[2], sHist[RUMI_HISTO_LENGTH], sKeep[RUMI_HISTO_LENGTH], sCount;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int kFree = 0x7FFFFFFF;
    if (A.nq > 0 && *A.overflow != 0) return;
    // ---- read once ----
    int cCnt[kResQ], cAsg[kResQ], cBlocks[kResQ];
    const uint32_t *cList[kResQ];
    uint32_t cHead[kResQ][kResK];
#pragma unroll
    for (int j = 0; j < kResQ; j++) {
        const int i = tid + j * nt;
        cCnt[j] = 0; cAsg[j] = -1; cBlocks[j] = 0; cList[j] = A.lists;
        if (i < A.nq) { cCnt[j] = A.counts[i]; cList[j] = A.lists + A.offsets[i]; cBlocks[j] = A.q[i].blocks; }
#pragma unroll
        for (int k = 0; k < kResK; k++) cHead[j][k] = k < cCnt[j] ? cList[j][k] : 0u;
    }
    for (int i = tid + kResQ * nt; i < A.nq; i += nt) A.assign[i] = -1;
    uint64_t taken0 = 0;                                                    // bit k: feature tid + k nt is taken before the call (nfeat <= 65536: list entries carry 16 bits)
    {
        int k = 0;
        for (int f = tid; f < A.nfeat; f += nt, k++) {
            bool t = false;
            if (A.featBlocked0) t = A.featBlocked0[f] != 0;
            else if (A.mode != MODE_BOW) { const int id = A.featMp[f]; t = id >= 0 && A.mpObs[id] > 0; }
            taken0 |= (uint64_t)t << k;
            blockedFrom[f] = t ? -1 : kFree;
        }
    }
    if (tid < RUMI_HISTO_LENGTH) sHist[tid] = 0;
    if (tid < 2) sChanged[tid] = 0;
    if (tid == 0) sCount = 0;
    auto pick_of = [&](int i, int cnt, const uint32_t *L, const uint32_t *head /* kResK entries in registers, or null */) -> int {
        if (cnt <= 0) return -1;
        int bestDist = 256, bestDist2 = 256, bestLevel = -1, bestLevel2 = -1, bestIdx = -1;
        const bool sortedList = cnt <= kSortMax;                           // then entries come in (distance, candidate order)
        bool done = false;
        auto take = [&](uint32_t e) {
            const int f = (int)(e & 0xFFFF);
            if (blockedFrom[f] < i) return;                                 // taken by an earlier query (or before the call)
            const int d = (int)((e >> 16) & 0x1FF), lv = (int)((e >> 25) & 15);
            if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestLevel2 = bestLevel; bestLevel = lv; bestIdx = f; }
            else if (d < bestDist2) { bestLevel2 = lv; bestDist2 = d; done = sortedList; }
            else done = sortedList;                                         // equal to the second best: nothing later can change either
        };
        int k = 0;
        if (head) {
#pragma unroll
            for (int h = 0; h < kResK; h++) if (h < cnt && !done) take(head[h]);
            k = kResK;
        }
        for (; k < cnt && !done; k++) take(L[k]);
        int pick = -1;
        if (A.mode == MODE_MAPPOINTS) {                                     // ORBmatcher.cc:106-111
            if (bestDist <= RUMI_TH_HIGH && !(bestLevel == bestLevel2 && (float)bestDist > A.nnratio * (float)bestDist2)) pick = bestIdx;
        } else if (A.mode == MODE_FRAME) {                                  // :1577
            if (bestDist <= RUMI_TH_HIGH) pick = bestIdx;
        } else if (A.mode == MODE_BOW) {                                    // :283-285
            if (bestDist <= RUMI_TH_LOW && (float)bestDist < A.nnratio * (float)bestDist2) pick = bestIdx;
        } else if (A.mode == MODE_BOW_KF) {                                 // :753-754
            if (bestDist < RUMI_TH_LOW && (float)bestDist < A.nnratio * (float)bestDist2) pick = bestIdx;
        } else if (A.mode == MODE_SIM3) {                                   // :463 / :571
            if ((float)bestDist <= A.thrF) pick = bestIdx;
        } else if (A.mode == MODE_FUSE) {                                   // Fuse :1161 / :1277 (TH_LOW), SearchBySim3 :1399 / :1475 (TH_HIGH)
            if (bestDist <= A.thrI) pick = bestIdx;
        } else {                                                            // MODE_RELOC :1757
            if (bestDist <= A.thrI) pick = bestIdx;
        }
        return pick;
    };
    for (int round = 0; round <= A.nq + 1; round++) {
        __syncthreads();                                                    // occupancy reset (below, or the initial one above) visible
        // occupancy as the previous round's assignments imply it
#pragma unroll
        for (int j = 0; j < kResQ; j++)
            if (cAsg[j] >= 0 && cBlocks[j]) atomicMin(&blockedFrom[cAsg[j]], tid + j * nt);
        for (int i = tid + kResQ * nt; i < A.nq; i += nt) {
            const int f = A.assign[i];
            if (f >= 0 && A.q[i].blocks) atomicMin(&blockedFrom[f], i);
        }
        __syncthreads();
        int changed = 0;
#pragma unroll
        for (int j = 0; j < kResQ; j++) {
            const int i = tid + j * nt;
            if (i >= A.nq) continue;
            const int pick = pick_of(i, cCnt[j], cList[j], cHead[j]);
            if (pick != cAsg[j]) { changed = 1; cAsg[j] = pick; }
        }
        for (int i = tid + kResQ * nt; i < A.nq; i += nt) {
            const int pick = pick_of(i, A.counts[i], A.lists + A.offsets[i], nullptr);
            if (pick != A.assign[i]) { changed = 1; A.assign[i] = pick; }
        }
        if (changed) sChanged[round & 1] = 1;
        __syncthreads();
        const int any = sChanged[round & 1];
        if (tid == 0) sChanged[(round + 1) & 1] = 0;                        // last read before this round's barriers, next written after the next round's
        if (!any) break;
        int k = 0;
        for (int f = tid; f < A.nfeat; f += nt, k++) blockedFrom[f] = ((taken0 >> k) & 1) ? -1 : kFree;
    }
#pragma unroll
    for (int j = 0; j < kResQ; j++) { const int i = tid + j * nt; if (i < A.nq) A.assign[i] = cAsg[j]; }
    // results: a feature keeps the LAST query that assigned it (later assignments overwrite, as in the loop)
    int32_t *last = blockedFrom;                                        // reuse LDS: last assigning query per feature
    for (int f = tid; f < A.nfeat; f += nt) last[f] = -1;
    __syncthreads();
    const bool useHist = A.checkOri && A.mode != MODE_MAPPOINTS && A.mode != MODE_SIM3;
    int local = 0;
    auto assigned = [&](int i, int j) { return j == 0 ? cAsg[0] : j == 1 ? cAsg[1] : A.assign[i]; };
    static_assert(kResQ == 2, "assigned() spells the register copies out");
    for (int i = tid, j = 0; i < A.nq; i += nt, j++) {
        const int f = assigned(i, j);
        if (f < 0) continue;
        local++;
        atomicMax(&last[f], i);
        if (useHist) atomicAdd(&sHist[rot_bin(A.q[i].angle, A.featKeys[f].angle)], 1);
    }
    if (local) atomicAdd(&sCount, local);
    __syncthreads();
    if (tid < 64) {                                                      // ComputeThreeMaxima, ORBmatcher.cc:1795-1826, by the lanes of one wave:
        // the scan with its strict comparisons keeps the three largest counts ordered by (count descending, bin ascending); empty bins never enter
        const int s = tid < RUMI_HISTO_LENGTH ? sHist[tid] : 0;
        int keep = 1, removed = 0;
        if (useHist) {
            uint32_t key = s > 0 ? ((uint32_t)s << 6) | (uint32_t)(63 - tid) : 0u;
            const uint32_t k1 = wave_max_u32(key);
            if (key == k1) key = 0;
            const uint32_t k2 = wave_max_u32(key);
            if (key == k2) key = 0;
            const uint32_t k3 = wave_max_u32(key);
            const int max1 = (int)(k1 >> 6), max2 = (int)(k2 >> 6), max3 = (int)(k3 >> 6);
            int ind1 = k1 ? 63 - (int)(k1 & 63) : -1, ind2 = k2 ? 63 - (int)(k2 & 63) : -1, ind3 = k3 ? 63 - (int)(k3 & 63) : -1;
            if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
            else if ((float)max3 < 0.1f * (float)max1) ind3 = -1;
            keep = (tid == ind1 || tid == ind2 || tid == ind3);
            removed = wave_sum_i32(keep ? 0 : s);
        }
        if (tid < RUMI_HISTO_LENGTH) sKeep[tid] = keep;
        if (tid == 0) *A.nmatches = sCount - removed;
    }
    __syncthreads();
    // MODE_BOW starts from an all-NULL vector (ORBmatcher.cc:201); the other modes update the frame's vector in place
    if (A.mode == MODE_BOW)
        for (int f = tid; f < A.nfeat; f += nt) A.featMp[f] = -1;
    __syncthreads();
    if (A.mode != MODE_BOW_KF)
        for (int f = tid; f < A.nfeat; f += nt)
            if (last[f] >= 0) A.featMp[f] = A.q[last[f]].mpId;
    __syncthreads();
    if (useHist)                                                        // entries of the rejected bins are set to NULL
        for (int i = tid, j = 0; i < A.nq; i += nt, j++) {
            const int f = assigned(i, j);
            if (f >= 0 && !sKeep[rot_bin(A.q[i].angle, A.featKeys[f].angle)]) {
                if (A.mode == MODE_BOW_KF) A.assign[i] = -1;           // SearchByBoW(KF,KF) reports per QUERY (vpMatches12[idx1])
                else A.featMp[f] = -1;
            }
        }
    if (A.gXw) {                                                        // (uniform) the Tracking step's next launch would be this gather
        __shared__ int sGatherWave[16], sGatherBase;
        __syncthreads();                                                // the vector is final
        gather_correspondences(A.nfeat, A.featKeys, A.featMp, A.gMpPos, A.gInvSigma2, A.gXw, A.gObs, A.gW, A.gIdx, A.gStart, A.gSnapshot, sGatherWave, &sGatherBase);
    }
}

// SearchForInitialization resolve (ORBmatcher.cc:593-679).  The skip rule `vMatchedDistance[i2] <= dist` makes every query depend
// on the best distance accepted so far for each candidate, so the queries are replayed IN ORDER by one wave; the lanes share the
// candidate list of the current query.  Called once per initialisation attempt (a few thousand queries): latency, not throughput.
struct InitArgs {
    int n1, n2;
    const Query *q;
    const int32_t *counts, *offsets;
    const uint32_t *lists;
    const RumiKeyPoint *keys2;
    int32_t *matches12;             // out [n1]
    float *prevMatched;             // in/out [n1][2]
    int32_t *nmatches;
    float nnratio;
    int checkOri;
    const int32_t *overflow;
};

__global__ __launch_bounds__(64) void k_resolve_init(InitArgs A) {
    extern __shared__ int32_t sInit[];          // matchedDist[n2] | matches21[n2]
    __shared__ int sHist[RUMI_HISTO_LENGTH], sKeep[RUMI_HISTO_LENGTH];
    int32_t *matchedDist = sInit, *matches21 = sInit + A.n2;
    const int lane = threadIdx.x;
    const int kInf = 0x7FFFFFFF;
    if (*A.overflow != 0) return;
    for (int f = lane; f < A.n2; f += 64) { matchedDist[f] = kInf; matches21[f] = -1; }
    for (int i = lane; i < A.n1; i += 64) A.matches12[i] = -1;
    if (lane < RUMI_HISTO_LENGTH) sHist[lane] = 0;
    __syncthreads();
    int nmatches = 0;
    for (int i1 = 0; i1 < A.n1; i1++) {
        const int cnt = A.counts[i1];
        if (cnt == 0) continue;
        const uint32_t *L = A.lists + A.offsets[i1];
        int b1 = kInf, b2 = kInf, bKey = kInf;            // best, second-best distance; best as dist<<16 | list position
        int bFeat = -1;
        for (int k = lane; k < cnt; k += 64) {
            const uint32_t e = L[k];
            const int f = (int)(e & 0xFFFF), d = (int)((e >> 16) & 0x1FF);
            if (matchedDist[f] <= d) continue;                            // :617
            if (d < b1) { b2 = b1; b1 = d; bKey = (d << 16) | k; bFeat = f; }
            else if (d < b2) b2 = d;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int c1 = __shfl_xor(b1, o), c2 = __shfl_xor(b2, o), cKey = __shfl_xor(bKey, o), cFeat = __shfl_xor(bFeat, o);
            b2 = min(max(b1, c1), min(b2, c2));
            b1 = min(b1, c1);
            if (cKey < bKey) { bKey = cKey; bFeat = cFeat; }
        }
        // :629-638 (uniform across the wave)
        if (b1 <= RUMI_TH_LOW && (float)b1 < (float)b2 * A.nnratio) {
            if (lane == 0) {
                const int prev = matches21[bFeat];
                if (prev >= 0) A.matches12[prev] = -1;
                A.matches12[i1] = bFeat;
                matches21[bFeat] = i1;
                matchedDist[bFeat] = b1;
                if (A.checkOri) sHist[rot_bin(A.q[i1].angle, A.keys2[bFeat].angle)]++;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    if (lane == 0) {
        for (int i = 0; i < RUMI_HISTO_LENGTH; i++) sKeep[i] = 1;
        if (A.checkOri) {                                                   // ComputeThreeMaxima over ALL accepted (also stolen) entries
            int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
            for (int i = 0; i < RUMI_HISTO_LENGTH; i++) {
                const int s = sHist[i];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
                else if (s > max3) { max3 = s; ind3 = i; }
            }
            if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
            else if ((float)max3 < 0.1f * (float)max1) ind3 = -1;
            for (int i = 0; i < RUMI_HISTO_LENGTH; i++) sKeep[i] = (i == ind1 || i == ind2 || i == ind3);
        }
    }
    __syncthreads();
    // a surviving match keeps the bin it was accepted with (its feature never changes afterwards): :661-677
    for (int i = lane; i < A.n1; i += 64) {
        int f = A.matches12[i];
        if (f >= 0 && A.checkOri && !sKeep[rot_bin(A.q[i].angle, A.keys2[f].angle)]) { A.matches12[i] = -1; f = -1; }
        if (f >= 0) { nmatches++; A.prevMatched[2 * i] = A.keys2[f].x; A.prevMatched[2 * i + 1] = A.keys2[f].y; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nmatches += __shfl_xor(nmatches, o);
    if (lane == 0) *A.nmatches = nmatches;
}

// SearchForTriangulation, monocular branch (ORBmatcher.cc:806-1013).  vbMatched2 is never set in the reference, so the queries
// are independent: per KF1 feature without a map point, the LAST candidate of minimal distance (<= TH_LOW, `dist > bestDist`
// rejects, so ties move to the later one) among those passing the epipole and epipolar tests.  One thread per node of KF1.
struct TriArgs {
    int nn1, nn2;
    const uint32_t *nodes1; const int32_t *off1; const uint32_t *idx1;
    const uint32_t *nodes2; const int32_t *off2; const uint32_t *idx2;
    const RumiKeyPoint *keys1, *keys2;
    const uint8_t *desc1, *desc2;
    const int32_t *mp1, *mp2;
    const float *scale2;            // KF2 mvScaleFactors
    const float *geom;              // F12 row-major [9], epipole [2]
    int coarse;
    int32_t *assign;                // [entries of fv1] chosen KF2 feature or -1
};

__global__ void k_tri_match(TriArgs A) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= A.nn1) return;
    int lo = 0, hi = A.nn2;
    const uint32_t id = A.nodes1[a];
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (A.nodes2[mid] < id) lo = mid + 1; else hi = mid; }
    const bool hit = lo < A.nn2 && A.nodes2[lo] == id;
    const float *F = A.geom;
    const float epx = A.geom[9], epy = A.geom[10];
    for (int p = A.off1[a]; p < A.off1[a + 1]; p++) {
        int best = -1;
        const int i1 = (int)A.idx1[p];
        if (hit && A.mp1[i1] < 0) {
            const RumiKeyPoint k1 = A.keys1[i1];
            uint32_t d1[8];
#pragma unroll
            for (int k = 0; k < 8; k++) d1[k] = reinterpret_cast<const uint32_t *>(A.desc1 + (size_t)i1 * 32)[k];
            // epipolar line l = x1' F12 (Pinhole.cpp:114-117)
            const float la = k1.x * F[0] + k1.y * F[3] + F[6];
            const float lb = k1.x * F[1] + k1.y * F[4] + F[7];
            const float lc = k1.x * F[2] + k1.y * F[5] + F[8];
            const float den = la * la + lb * lb;
            int bestDist = RUMI_TH_LOW;
            for (int c = A.off2[lo]; c < A.off2[lo + 1]; c++) {
                const int i2 = (int)A.idx2[c];
                if (A.mp2[i2] >= 0) continue;
                const int dist = hamming256(d1, reinterpret_cast<const uint32_t *>(A.desc2 + (size_t)i2 * 32));
                if (dist > RUMI_TH_LOW || dist > bestDist) continue;
                const RumiKeyPoint k2 = A.keys2[i2];
                const float ex = epx - k2.x, ey = epy - k2.y;
                if (ex * ex + ey * ey < 100 * A.scale2[k2.octave]) continue;                       // :912-918
                if (!A.coarse) {
                    const float num = la * k2.x + lb * k2.y + lc;
                    if (den == 0) continue;
                    const float dsqr = num * num / den;
                    const float s2 = A.scale2[k2.octave] * A.scale2[k2.octave];                    // mvLevelSigma2
                    if (!((double)dsqr < 3.84 * (double)s2)) continue;
                }
                best = i2; bestDist = dist;
            }
        }
        A.assign[p] = best;
    }
}

// rotation-histogram filter + match count for k_tri_match (ORBmatcher.cc:964-1001); single workgroup
__global__ __launch_bounds__(256) void k_tri_filter(int nq, const uint32_t *idx1, const RumiKeyPoint *keys1, const RumiKeyPoint *keys2,
                                                    int32_t *assign, int checkOri, int32_t *nmatches) {
    __shared__ int sHist[RUMI_HISTO_LENGTH], sKeep[RUMI_HISTO_LENGTH], sCount;
    const int tid = threadIdx.x;
    if (tid < RUMI_HISTO_LENGTH) { sHist[tid] = 0; sKeep[tid] = 1; }
    if (tid == 0) sCount = 0;
    __syncthreads();
    if (checkOri) {
        for (int p = tid; p < nq; p += 256) {
            const int f = assign[p];
            if (f >= 0) atomicAdd(&sHist[rot_bin(keys1[idx1[p]].angle, keys2[f].angle)], 1);
        }
        __syncthreads();
        if (tid == 0) {
            int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
            for (int i = 0; i < RUMI_HISTO_LENGTH; i++) {
                const int s = sHist[i];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
                else if (s > max3) { max3 = s; ind3 = i; }
            }
            if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
            else if ((float)max3 < 0.1f * (float)max1) ind3 = -1;
            for (int i = 0; i < RUMI_HISTO_LENGTH; i++) sKeep[i] = (i == ind1 || i == ind2 || i == ind3);
        }
        __syncthreads();
    }
    int local = 0;
    for (int p = tid; p < nq; p += 256) {
        const int f = assign[p];
        if (f < 0) continue;
        if (checkOri && !sKeep[rot_bin(keys1[idx1[p]].angle, keys2[f].angle)]) assign[p] = -1;
        else local++;
    }
    atomicAdd(&sCount, local);
    __syncthreads();
    if (tid == 0) *nmatches = sCount;
}

// ---- brute force ----------------------------------------------------------------------------------------------------
// grid (ceil(cap/256), B); 256 queries per workgroup in registers; train descriptors staged 256 at a time in LDS and
// read as broadcasts (every lane reads the same address: conflict-free).
// popcount(x) + acc in one instruction (the compiler re-associates a sum of popcounts into popcounts + 3-input adds)
__device__ __forceinline__ uint32_t popc_acc(uint32_t x, uint32_t acc) {
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__global__ __launch_bounds__(256) void k_bruteforce(const uint8_t *__restrict__ qd, const int32_t *__restrict__ nqArr,
                                                    const uint8_t *__restrict__ td, const int32_t *__restrict__ ntArr,
                                                    int countStride, long long qStride, long long tStride, int cap, int32_t *__restrict__ bestIdx,
                                                    int32_t *__restrict__ bestDist, int32_t *__restrict__ secondDist, int ring) {
    __shared__ uint4 tile[256 * 2];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int tb = ring > 0 ? (b + 1 == ring ? 0 : b + 1) : b;       // ring: frame b against its successor in the same buffer, the last against the first
    const int nq = min(nqArr[(size_t)b * countStride], cap), nt = min(ntArr[(size_t)tb * countStride], cap);
    const int qi = blockIdx.x * 256 + tid;
    if (blockIdx.x * 256 >= nq) return;
    uint32_t q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (qi < nq) {
        const uint4 *src = reinterpret_cast<const uint4 *>(qd + (size_t)b * qStride + (size_t)qi * 32);
        const uint4 a = src[0], c = src[1];
        q[0] = a.x; q[1] = a.y; q[2] = a.z; q[3] = a.w; q[4] = c.x; q[5] = c.y; q[6] = c.z; q[7] = c.w;
    }
    // best and second best as packed keys (distance << 16 | train index): "first minimum wins, a tie goes to the second place" is then
    // best = min(best, key), second = med3(best, second, key) — three instructions per pair next to the 8 xor + 8 popcount-accumulate
    uint32_t best = 256u << 16, second = (256u << 16) | 0xFFFFu;
    for (int t0 = 0; t0 < nt; t0 += 256) {
        const int m = min(256, nt - t0);
        __syncthreads();
        if (tid < m) {
            const uint4 *src = reinterpret_cast<const uint4 *>(td + (size_t)tb * tStride + (size_t)(t0 + tid) * 32);
            tile[tid * 2] = src[0];
            tile[tid * 2 + 1] = src[1];
        }
        __syncthreads();
        auto step = [&](int j) {
            const uint4 a = tile[j * 2], c = tile[j * 2 + 1];
            uint32_t d = popc_acc(q[0] ^ a.x, 0u);
            d = popc_acc(q[1] ^ a.y, d); d = popc_acc(q[2] ^ a.z, d); d = popc_acc(q[3] ^ a.w, d);
            d = popc_acc(q[4] ^ c.x, d); d = popc_acc(q[5] ^ c.y, d); d = popc_acc(q[6] ^ c.z, d); d = popc_acc(q[7] ^ c.w, d);
            const uint32_t key = (d << 16) | (uint32_t)(t0 + j);
            second = umed3(best, second, key);
            best = min(best, key);
        };
        int j = 0;
        for (; j + 4 <= m; j += 4) { step(j); step(j + 1); step(j + 2); step(j + 3); }   // four independent popcount chains in flight
        for (; j < m; j++) step(j);
    }
    if (qi < nq) {
        const size_t o = (size_t)b * cap + qi;
        const int b1 = (int)(best >> 16);
        bestIdx[o] = b1 < 256 ? (int)(best & 0xFFFFu) : -1;     // a distance of 256 never beats the initial 256 (the reference's strict <)
        bestDist[o] = b1; secondDist[o] = (int)(second >> 16);
    }
}


// ------------------------------------------------------------------------------------------------
// SearchByBoW(KF_k, F) for K candidate key-frames against ONE frame in one launch (Tracking::Relocalization walks the candidates of
// KeyFrameDatabase::DetectRelocalizationCandidates one by one, Tracking.cc:3240-3260; every walk starts from an empty vpMapPointMatches, so the
// K searches are independent).  Within one search a frame feature is taken by the first query that wins it -- but a frame feature lies in
// exactly ONE FeatureVector node, so that dependency never leaves a node: nodes run in parallel, the (few) key-frame features of a node
// sequentially.  16 lanes per (key-frame, node): the lanes share out the frame's features of the node, compute their Hamming distances
// to the current key-frame feature in parallel and reduce (best, second) with the reference's tie rules (ORBmatcher.cc:252-289).
// ------------------------------------------------------------------------------------------------
struct BowKF { int32_t n, nn, angle, desc, mp, good, nodes, off, idx, pad; };      // sizes and dword offsets of one key-frame's arrays in the block
struct BowBatch {
    const uint32_t *blk;          // the uploaded block (dword view)
    int K, nf, nnF;
    int fAngle, fDesc, fNodes, fOff, fIdx, kfTable;     // dword offsets
    int32_t *matches;             // [K][nf]  map-point index (per key-frame numbering), -1 none
    int8_t *rotBin;               // [K][nf]
    int32_t *hist;                // [K][32]
    int32_t *nmatch;              // [K]
    int32_t *err;                 // bit 0: a node with more than 512 frame features
    float nnratio;
    int checkOri;
};

__global__ __launch_bounds__(256) void k_bow_batch_match(BowBatch B) {
    const int k = blockIdx.y, lane = threadIdx.x & 15, a = blockIdx.x * 16 + (threadIdx.x >> 4);
    const BowKF *T = reinterpret_cast<const BowKF *>(B.blk + B.kfTable) + k;
    if (a >= T->nn) return;
    const uint32_t *kfNodes = B.blk + T->nodes, *fNodes = B.blk + B.fNodes;
    const uint32_t node = kfNodes[a];
    int lo = 0, hi = B.nnF;                                  // first frame node >= node (std::map order: ascending ids)
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (fNodes[mid] < node) lo = mid + 1; else hi = mid; }
    if (lo >= B.nnF || fNodes[lo] != node) return;
    const int32_t *fOff = reinterpret_cast<const int32_t *>(B.blk + B.fOff), *kOff = reinterpret_cast<const int32_t *>(B.blk + T->off);
    const int c0 = fOff[lo], nc = fOff[lo + 1] - c0;
    if (nc > 512) { if (lane == 0) atomicOr(B.err, 1); return; }
    const uint32_t *fIdx = B.blk + B.fIdx + c0, *kIdx = B.blk + T->idx;
    const uint32_t *fDesc = B.blk + B.fDesc, *kDesc = B.blk + T->desc;
    const float *fAngle = reinterpret_cast<const float *>(B.blk + B.fAngle), *kAngle = reinterpret_cast<const float *>(B.blk + T->angle);
    const int32_t *kMp = reinterpret_cast<const int32_t *>(B.blk + T->mp);
    const uint8_t *kGood = reinterpret_cast<const uint8_t *>(B.blk + T->good);
    uint32_t taken = 0;                                      // bit j: my candidate lane + 16 j already holds a map point
    for (int p = kOff[a]; p < kOff[a + 1]; p++) {
        const int iKF = (int)kIdx[p];
        if (!kGood[iKF]) continue;                           // no map point, or a bad one (:238-243)
        uint32_t q[8];
#pragma unroll
        for (int w = 0; w < 8; w++) q[w] = kDesc[(size_t)iKF * 8 + w];
        uint32_t best = (256u << 16) | 0xFFFFu, second = 256u;
        for (int j = 0, pos = lane; pos < nc; j++, pos += 16) {
            if ((taken >> j) & 1u) continue;
            const uint32_t *d = fDesc + (size_t)fIdx[pos] * 8;
            uint32_t dist = 0;
#pragma unroll
            for (int w = 0; w < 8; w++) dist += __popc(q[w] ^ d[w]);
            const uint32_t key = (dist << 16) | (uint32_t)pos;
            if (key < best) { second = best >> 16; best = key; }       // a strictly smaller distance, or the same at an earlier position
            else if (dist < second) second = dist;
        }
        // 16-lane reduction: best = smallest key; second = second smallest distance of the union
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            const uint32_t ob = __shfl_xor(best, o, 16), os = __shfl_xor(second, o, 16);
            const uint32_t loser = max(best, ob) >> 16;
            best = min(best, ob);
            second = min(min(second, os), loser);
        }
        const int bestDist1 = (int)(best >> 16), bestDist2 = (int)second;
        if (bestDist1 <= RUMI_TH_LOW && (float)bestDist1 < B.nnratio * (float)bestDist2) {
            const int pos = (int)(best & 0xFFFFu), f = (int)fIdx[pos];
            if (lane == (pos & 15)) taken |= 1u << (pos >> 4);
            if (lane == 0) {
                B.matches[(size_t)k * B.nf + f] = kMp[iKF];
                if (B.checkOri) {
                    const int bin = rot_bin(kAngle[iKF], fAngle[f]);
                    B.rotBin[(size_t)k * B.nf + f] = (int8_t)bin;
                    atomicAdd(&B.hist[k * 32 + bin], 1);
                }
            }
        }
    }
}

// rotation-histogram filter (ComputeThreeMaxima, ORBmatcher.cc:1795-1826) and the match count of every key-frame
__global__ __launch_bounds__(256) void k_bow_batch_finish(BowBatch B) {
    __shared__ int sKeep[RUMI_HISTO_LENGTH], sCount;
    const int k = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        sCount = 0;
        for (int i = 0; i < RUMI_HISTO_LENGTH; i++) sKeep[i] = 1;
        if (B.checkOri) {
            int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
            for (int i = 0; i < RUMI_HISTO_LENGTH; i++) {
                const int s = B.hist[k * 32 + i];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
                else if (s > max3) { max3 = s; ind3 = i; }
            }
            if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
            else if (max3 < 0.1f * (float)max1) ind3 = -1;
            for (int i = 0; i < RUMI_HISTO_LENGTH; i++) sKeep[i] = (i == ind1 || i == ind2 || i == ind3);
        }
    }
    __syncthreads();
    int local = 0;
    for (int f = tid; f < B.nf; f += 256) {
        int32_t &m = B.matches[(size_t)k * B.nf + f];
        if (m < 0) continue;
        if (B.checkOri && !sKeep[B.rotBin[(size_t)k * B.nf + f]]) m = -1; else local++;
    }
    atomicAdd(&sCount, local);
    __syncthreads();
    if (tid == 0) B.nmatch[k] = sCount;
}

}  // namespace rumi

using namespace rumi;

// ================================================ host side =======================================================
struct RumiMatcher {
    int device = 0, maxFeat = 0, maxQ = 0;
    size_t listCap = 0;
    // frame (train) side
    RumiKeyPoint *dKeys = nullptr; uint8_t *dDesc = nullptr; float *dScale = nullptr;
    uint16_t *dSorted = nullptr; int32_t *dCellStart = nullptr;
    uint32_t *dFvIdx = nullptr;      // frame FeatureVector indices (BoW)
    // query side
    Query *dQ = nullptr; uint8_t *dQDesc = nullptr; int32_t *dCounts = nullptr, *dOffsets = nullptr;
    uint32_t *dLists = nullptr;
    // results, one block so that one copy brings them back: [nmatches, list overflow, -, -][featMp maxFeat][assign maxQ]
    int32_t *dOut = nullptr, *hOut = nullptr;
    int32_t *dNmatches = nullptr, *dOverflow = nullptr, *dFeatMp = nullptr, *dAssign = nullptr;     // views into dOut
    // raw inputs of the query builders
    uint8_t *dU8a = nullptr, *dU8b = nullptr; float *dF[6] = {nullptr}; int32_t *dI[4] = {nullptr};
    RumiKeyPoint *dQKeys = nullptr; uint32_t *dNodesA = nullptr, *dNodesB = nullptr, *dIdxA = nullptr;
    int32_t *dOffA = nullptr, *dOffB = nullptr;
    float *dPose = nullptr;
    // uploads of one call: packed into a pinned block, copied once, scattered on the device (k_scatter)
    uint8_t *hStage = nullptr, *dStage = nullptr;
    size_t stageCap = 0, stageUsed = 0;
    int nseg = 0;
    // rumi_search_by_bow_batch: one pinned block up, one result block back (grown on demand)
    uint8_t *hBow = nullptr, *dBow = nullptr; size_t bowCap = 0;
    uint8_t *hBowOut = nullptr, *dBowOut = nullptr; size_t bowOutCap = 0;
    // k_grid of the uploaded frame, launched by flush_uploads once the key-points are in place
    const int32_t *gridNDev = nullptr;     // k_grid reads the count from the device (one call only: cleared by the flush)
    hipStream_t upStream = nullptr;        // where the next flush queues its copy and scatter (the caller orders its kernels behind them)
    bool gridPending = false; int gridN = 0; float gridMinX = 0, gridMinY = 0, gridWInv = 0, gridHInv = 0;
    const RumiKeyPoint *gridKeys = nullptr;      // key-points k_grid reads: dKeys, or a frame that already lies on the device (rumi_track_frame)
};
constexpr size_t kStageHeader = kMaxSegments * sizeof(Segment);

extern "C" int rumi_descriptor_distance(const uint8_t *a, const uint8_t *b) {
    uint64_t x[4], y[4];
    std::memcpy(x, a, 32); std::memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) + __builtin_popcountll(x[2] ^ y[2]) +
           __builtin_popcountll(x[3] ^ y[3]);
}

extern "C" void rumi_match_destroy(RumiMatcher *m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    void *p[] = {m->dKeys, m->dDesc, m->dScale, m->dSorted, m->dCellStart, m->dFvIdx, m->dQ, m->dQDesc, m->dCounts,
                 m->dOffsets, m->dLists, m->dOut, m->dU8a, m->dU8b, m->dF[0], m->dF[1], m->dF[2], m->dF[3],
                 m->dF[4], m->dF[5], m->dI[0], m->dI[1], m->dI[2], m->dI[3], m->dQKeys, m->dNodesA, m->dNodesB, m->dIdxA,
                 m->dOffA, m->dOffB, m->dPose, m->dStage};
    for (void *q : p) if (q) (void)hipFree(q);
    if (m->hStage) (void)hipHostFree(m->hStage);
    if (m->hOut) (void)hipHostFree(m->hOut);
    if (m->hBow) (void)hipHostFree(m->hBow);
    if (m->hBowOut) (void)hipHostFree(m->hBowOut);
    if (m->dBow) (void)hipFree(m->dBow);
    if (m->dBowOut) (void)hipFree(m->dBowOut);
    delete m;
}

template <class T> static int dalloc(T **p, size_t n) {
    *p = nullptr;
    HIP_TRY(hipMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(T)));
    return RUMI_OK;
}

extern "C" int rumi_match_create(int32_t max_features, int32_t max_queries, int32_t device, RumiMatcher **out) {
    if (!out) return RUMI_E_INVALID;
    *out = nullptr;
    if (max_features < 1 || max_features > kMaxSortN || max_queries < 1) {
        g_lastError = "rumi_match_create: max_features must be in 1..16384, max_queries >= 1";
        return RUMI_E_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        g_lastError = "no HIP device visible: librumi_hip has no CPU fallback";
        return RUMI_E_NO_DEVICE;
    }
    RumiMatcher *m = new RumiMatcher();
    if (device >= 0) m->device = device; else if (hipGetDevice(&m->device) != hipSuccess) m->device = 0;
    if (hipSetDevice(m->device) != hipSuccess) { delete m; return RUMI_E_NO_DEVICE; }
    m->maxFeat = max_features; m->maxQ = max_queries;
    m->listCap = (size_t)max_queries * 256 + 65536;     // grown on demand
    const size_t F = max_features, Q = max_queries;
    int rc;
#define TRYA(x) if ((rc = (x)) != RUMI_OK) { rumi_match_destroy(m); return rc; }
    TRYA(dalloc(&m->dKeys, F)); TRYA(dalloc(&m->dDesc, F * 32)); TRYA(dalloc(&m->dScale, 64));
    TRYA(dalloc(&m->dSorted, F)); TRYA(dalloc(&m->dCellStart, kGridCells + 2));
    TRYA(dalloc(&m->dFvIdx, F));
    TRYA(dalloc(&m->dQ, Q)); TRYA(dalloc(&m->dQDesc, Q * 32)); TRYA(dalloc(&m->dCounts, Q + 1)); TRYA(dalloc(&m->dOffsets, Q + 1));
    TRYA(dalloc(&m->dLists, m->listCap));
    TRYA(dalloc(&m->dOut, 4 + F + Q));
    m->dNmatches = m->dOut; m->dOverflow = m->dOut + 1; m->dFeatMp = m->dOut + 4; m->dAssign = m->dOut + 4 + F;
    TRYA(dalloc(&m->dU8a, Q)); TRYA(dalloc(&m->dU8b, std::max(Q, F)));
    for (auto &f : m->dF) TRYA(dalloc(&f, Q * 3));
    for (auto &i : m->dI) TRYA(dalloc(&i, Q + 1));
    TRYA(dalloc(&m->dQKeys, Q)); TRYA(dalloc(&m->dNodesA, Q)); TRYA(dalloc(&m->dNodesB, F)); TRYA(dalloc(&m->dIdxA, Q));
    TRYA(dalloc(&m->dOffA, Q + 1)); TRYA(dalloc(&m->dOffB, F + 1)); TRYA(dalloc(&m->dPose, 32));
    m->stageCap = kStageHeader + F * 112 + Q * 224 + 65536;
    TRYA(dalloc(&m->dStage, m->stageCap));
    if (hipHostMalloc((void **)&m->hStage, m->stageCap, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&m->hOut, (4 + F + Q) * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) {
        g_lastError = "rumi_match_create: pinned host allocation failed";
        rumi_match_destroy(m);
        return RUMI_E_NO_DEVICE;
    }
    m->stageUsed = kStageHeader;
#undef TRYA
    *out = m;
    return RUMI_OK;
}

// Queue `bytes` of host data for the array `dst`; nothing moves until flush_uploads.
static int stage_add(RumiMatcher *m, void *dst, const void *src, size_t bytes) {
    if (bytes == 0) return RUMI_OK;
    const size_t off = (m->stageUsed + 15) & ~(size_t)15;
    if (m->nseg >= kMaxSegments || off + bytes > m->stageCap) {
        g_lastError = "matcher upload block exhausted (raise max_features / max_queries)";
        return RUMI_E_CAPACITY;
    }
    std::memcpy(m->hStage + off, src, bytes);
    reinterpret_cast<Segment *>(m->hStage)[m->nseg++] = Segment{dst, (uint32_t)off, (uint32_t)bytes};
    m->stageUsed = off + bytes;
    return RUMI_OK;
}
#define H2D(dst, src, n) do { const int rcS_ = stage_add(m, (dst), (src), (size_t)(n) * sizeof(*(dst))); if (rcS_ != RUMI_OK) return rcS_; } while (0)

// One host-to-device copy for everything queued, the scatter, then the grid of the uploaded frame.
static int flush_uploads(RumiMatcher *m) {
    if (m->nseg > 0) {
        HIP_TRY(hipMemcpyAsync(m->dStage, m->hStage, m->stageUsed, hipMemcpyHostToDevice, m->upStream));
        hipLaunchKernelGGL(k_scatter, dim3(8, m->nseg), dim3(256), 0, m->upStream, m->dStage, m->nseg);
        m->nseg = 0;
        m->stageUsed = kStageHeader;
    }
    if (m->gridPending) {
        hipLaunchKernelGGL(k_grid, dim3(1), dim3(1024), 0, nullptr, m->gridN, m->gridKeys ? m->gridKeys : m->dKeys, m->gridMinX, m->gridMinY, m->gridWInv, m->gridHInv,
                           m->dSorted, m->dCellStart, m->gridNDev);
        m->gridPending = false;
        m->gridNDev = nullptr;
    }
    return RUMI_OK;
}
#define FLUSH(m) do { const int rcF_ = flush_uploads(m); if (rcF_ != RUMI_OK) return rcF_; } while (0)

// A call that fails between stage_add and flush must not leak its queue into the next one.
static void reset_uploads(RumiMatcher *m) { m->nseg = 0; m->stageUsed = kStageHeader; m->gridPending = false; }

static int upload_frame(RumiMatcher *m, const RumiFrameFeatures *F, FrameDev *fd) {
    reset_uploads(m);
    if (!F || F->n < 0 || F->n > m->maxFeat || F->nlevels < 1 || F->nlevels > 64 || !(F->max_x > F->min_x) || !(F->max_y > F->min_y)) {
        g_lastError = "bad RumiFrameFeatures (n, nlevels or bounds)";
        return RUMI_E_INVALID;
    }
    if (F->n > 0) { H2D(m->dKeys, F->keys_un, F->n); H2D(m->dDesc, F->desc, (size_t)F->n * 32); }
    H2D(m->dScale, F->scale_factors, F->nlevels);
    static const int32_t kZeroHeader[4] = {0, 0, 0, 0};     // result header [nmatches | list overflow | - | -]: cleared by the same scatter
    H2D(m->dOut, kZeroHeader, 4);
    fd->n = F->n; fd->keys = m->dKeys; fd->desc = m->dDesc;
    fd->minX = F->min_x; fd->minY = F->min_y; fd->maxX = F->max_x; fd->maxY = F->max_y;
    fd->wInv = (float)kGridCols / (float)(F->max_x - F->min_x);     // Frame.cc:322-323
    fd->hInv = (float)kGridRows / (float)(F->max_y - F->min_y);
    fd->sortedIdx = m->dSorted; fd->cellStart = m->dCellStart; fd->scale = m->dScale;
    m->gridPending = true; m->gridKeys = nullptr; m->gridN = F->n; m->gridMinX = fd->minX; m->gridMinY = fd->minY; m->gridWInv = fd->wInv; m->gridHInv = fd->hInv;
    return RUMI_OK;
}

// count pass, scan, fill pass.  The fill pass refuses to write past the list arena and raises the overflow word instead; the
// caller sees it in the result block, grows the arena and repeats the call (run_search) — no mid-pipeline read-back.
static int build_lists(RumiMatcher *m, int mode, int nq, const FrameDev &fd, const uint8_t *dQueryDesc, bool retry, bool fused) {
    FLUSH(m);
    if (retry) HIP_TRY(hipMemsetAsync(m->dOut, 0, 4 * sizeof(int32_t), nullptr));   // the first attempt's header was cleared with the frame upload
    if (nq > 0 && fused) {
        // one launch: every query's list in a fixed slot of the arena (k_candidates<2>)
        const int slot = (int)std::min<size_t>(m->listCap / (size_t)nq, 0x7FFFFFFF / (size_t)nq);
        hipLaunchKernelGGL(k_candidates<2>, dim3((nq + 3) / 4), dim3(256), 0, nullptr, mode, nq, m->dQ, fd, dQueryDesc, m->dFvIdx,
                           m->dCounts, m->dOffsets, m->dLists, slot, m->dOverflow);
    } else if (nq > 0) {
        hipLaunchKernelGGL(k_candidates<0>, dim3((nq + 3) / 4), dim3(256), 0, nullptr, mode, nq, m->dQ, fd, dQueryDesc, m->dFvIdx,
                           m->dCounts, m->dOffsets, m->dLists, 0, m->dOverflow);
        hipLaunchKernelGGL(k_scan, dim3(1), dim3(256), 0, nullptr, nq, m->dCounts, m->dOffsets);
        hipLaunchKernelGGL(k_candidates<1>, dim3((nq + 3) / 4), dim3(256), 0, nullptr, mode, nq, m->dQ, fd, dQueryDesc, m->dFvIdx,
                           m->dCounts, m->dOffsets, m->dLists, (int)std::min<size_t>(m->listCap, 0x7FFFFFFF), m->dOverflow);
    }
    return RUMI_OK;
}

// bring back [header | featMp | assign] with one copy; returns RUMI_E_CAPACITY-like signal through *overflowTotal
static int fetch_results(RumiMatcher *m, int nfeat, int nq, bool wantAssign) {
    const size_t ints = wantAssign ? (size_t)4 + m->maxFeat + std::max(nq, 0) : (size_t)4 + std::max(nfeat, 0);
    HIP_TRY(hipMemcpy(m->hOut, m->dOut, ints * sizeof(int32_t), hipMemcpyDeviceToHost));
    return RUMI_OK;
}

static int grow_lists(RumiMatcher *m, size_t need) {
    (void)hipFree(m->dLists);
    m->dLists = nullptr;
    m->listCap = need * 2;
    return dalloc(&m->dLists, m->listCap);
}

// candidate lists, then the fix-point resolve
static int run_search(RumiMatcher *m, int mode, int nq, const FrameDev &fd, const uint8_t *dQueryDesc, const int32_t *dMpObs,
                      float nnratio, int checkOri, int32_t *hostFeatMp, int32_t *nmatchesOut, const uint8_t *dBlocked0 = nullptr,
                      float thrF = 0.f, int thrI = 0, int32_t *hostAssign = nullptr) {
    // first with every query's list in a fixed slot (one candidate launch); a query that does not fit falls back to count / scan / fill, which
    // sizes the lists exactly and grows the arena when needed
    static const bool noFused = std::getenv("RUMI_MATCH_NO_FUSED") != nullptr;
    bool fused = !noFused && nq > 0 && m->listCap / (size_t)nq >= 64;
    for (int attempt = 0, grown = 0; attempt < 4; attempt++) {
        const int rcl = build_lists(m, mode, nq, fd, dQueryDesc, attempt > 0, fused);
        if (rcl != RUMI_OK) return rcl;
        ResolveArgs A{mode, nq, fd.n, m->dQ, m->dCounts, m->dOffsets, m->dLists, fd.keys, dMpObs, m->dFeatMp, m->dAssign, m->dNmatches,
                      nnratio, checkOri, dBlocked0, thrF, thrI, m->dOverflow};
        hipLaunchKernelGGL(k_resolve, dim3(1), dim3(1024), (size_t)std::max(fd.n, 1) * sizeof(int32_t), nullptr, A);
        HIP_TRY(hipGetLastError());
        const int rcf = fetch_results(m, fd.n, nq, hostAssign != nullptr);
        if (rcf != RUMI_OK) return rcf;
        if (m->hOut[1] == 0) break;
        if (m->hOut[1] == kFusedOverflow) { fused = false; continue; }      // the resolve did not run; repeat with exact list sizes
        // list arena too small: the resolve did not run and the frame's map-point vector is untouched
        if (grown) { g_lastError = "candidate list arena overflow after growing"; return RUMI_E_CAPACITY; }
        const int rcg = grow_lists(m, (size_t)m->hOut[1]);
        if (rcg != RUMI_OK) return rcg;
        grown = 1;
    }
    *nmatchesOut = m->hOut[0];
    if (fd.n > 0 && hostFeatMp) std::memcpy(hostFeatMp, m->hOut + 4, (size_t)fd.n * sizeof(int32_t));
    if (nq > 0 && hostAssign) std::memcpy(hostAssign, m->hOut + 4 + m->maxFeat, (size_t)nq * sizeof(int32_t));
    return RUMI_OK;
}

extern "C" int rumi_search_by_projection_mappoints(RumiMatcher *m, const RumiFrameFeatures *F, int32_t nmp,
                                                   const uint8_t *track_in_view, const float *proj_x, const float *proj_y,
                                                   const int32_t *scale_level, const float *view_cos, const float *track_depth,
                                                   const uint8_t *is_bad, const uint8_t *mp_desc, const int32_t *mp_obs, float th,
                                                   int32_t far_points, float th_far_points, float nnratio, int32_t *frame_mp,
                                                   int32_t *nmatches_out) {
    if (!m || !nmatches_out || !frame_mp || nmp < 0) return RUMI_E_INVALID;
    if (nmp > m->maxQ) { g_lastError = "more map points than max_queries"; return RUMI_E_CAPACITY; }
    HIP_TRY(hipSetDevice(m->device));
    FrameDev fd;
    int rc = upload_frame(m, F, &fd);
    if (rc != RUMI_OK) return rc;
    if (F->n > 0) H2D(m->dFeatMp, frame_mp, F->n);
    if (nmp > 0) {
        H2D(m->dU8a, track_in_view, nmp); H2D(m->dU8b, is_bad, nmp);
        H2D(m->dF[0], proj_x, nmp); H2D(m->dF[1], proj_y, nmp); H2D(m->dF[2], view_cos, nmp); H2D(m->dF[3], track_depth, nmp);
        H2D(m->dI[0], scale_level, nmp); H2D(m->dI[1], mp_obs, nmp);
        H2D(m->dQDesc, mp_desc, (size_t)nmp * 32);
        FLUSH(m);
        hipLaunchKernelGGL(k_queries_mappoints, dim3((nmp + 255) / 256), dim3(256), 0, nullptr, nmp, m->dU8a, m->dF[0], m->dF[1],
                           m->dI[0], m->dF[2], m->dF[3], m->dU8b, m->dI[1], m->dScale, th, far_points, th_far_points, m->dQ);
    }
    return run_search(m, MODE_MAPPOINTS, nmp, fd, m->dQDesc, m->dI[1], nnratio, 0, frame_mp, nmatches_out);
}

extern "C" int rumi_search_by_projection_frame(RumiMatcher *m, const RumiFrameFeatures *Cur, const float *Tcw7, const float *K4,
                                               const RumiKeyPoint *last_keys, int32_t nlast, const int32_t *last_mp,
                                               const uint8_t *last_outlier, int32_t nmp, const float *mp_pos, const uint8_t *mp_desc,
                                               const int32_t *mp_obs, float th, int32_t check_orientation, int32_t *cur_mp,
                                               int32_t *nmatches_out) {
    if (!m || !nmatches_out || !cur_mp || nlast < 0 || nmp < 0 || !Tcw7 || !K4) return RUMI_E_INVALID;
    if (nlast > m->maxQ || nmp > m->maxQ) { g_lastError = "more last-frame features / map points than max_queries"; return RUMI_E_CAPACITY; }
    HIP_TRY(hipSetDevice(m->device));
    FrameDev fd;
    int rc = upload_frame(m, Cur, &fd);
    if (rc != RUMI_OK) return rc;
    if (Cur->n > 0) H2D(m->dFeatMp, cur_mp, Cur->n);
    float pose[11];
    std::memcpy(pose, Tcw7, 7 * sizeof(float)); std::memcpy(pose + 7, K4, 4 * sizeof(float));
    H2D(m->dPose, pose, 11);
    if (nmp > 0) { H2D(m->dF[0], mp_pos, (size_t)nmp * 3); H2D(m->dI[1], mp_obs, nmp); H2D(m->dQDesc, mp_desc, (size_t)nmp * 32); }
    if (nlast > 0) {
        H2D(m->dQKeys, last_keys, nlast); H2D(m->dI[0], last_mp, nlast); H2D(m->dU8a, last_outlier, nlast);
        FLUSH(m);
        hipLaunchKernelGGL(k_queries_frame, dim3((nlast + 255) / 256), dim3(256), 0, nullptr, nlast, m->dQKeys, m->dI[0], m->dU8a,
                           m->dF[0], m->dI[1], m->dPose, m->dPose + 7, m->dScale, th, fd.minX, fd.minY, fd.maxX, fd.maxY, m->dQ);
    }
    return run_search(m, MODE_FRAME, nlast, fd, m->dQDesc, m->dI[1], 0.f, check_orientation, cur_mp, nmatches_out);
}

extern "C" int rumi_search_by_bow(RumiMatcher *m, const RumiFrameFeatures *KF, const RumiFeatureVector *kf_fv, const int32_t *kf_mp,
                                  int32_t nmp, const uint8_t *mp_bad, const RumiFrameFeatures *F, const RumiFeatureVector *f_fv,
                                  float nnratio, int32_t check_orientation, int32_t *matches, int32_t *nmatches_out) {
    if (!m || !KF || !kf_fv || !f_fv || !matches || !nmatches_out || nmp < 0 || !kf_mp) return RUMI_E_INVALID;
    const int nqe = kf_fv->n_nodes > 0 ? kf_fv->offsets[kf_fv->n_nodes] : 0;     // one query per FeatureVector entry
    const int nfe = f_fv->n_nodes > 0 ? f_fv->offsets[f_fv->n_nodes] : 0;
    if (KF->n > m->maxQ || nqe > m->maxQ || nmp > m->maxQ || kf_fv->n_nodes > m->maxQ || nfe > m->maxFeat || f_fv->n_nodes > m->maxFeat) {
        g_lastError = "SearchByBoW: sizes exceed the matcher's capacities";
        return RUMI_E_CAPACITY;
    }
    HIP_TRY(hipSetDevice(m->device));
    FrameDev fd;
    int rc = upload_frame(m, F, &fd);
    if (rc != RUMI_OK) return rc;
    if (KF->n > 0) { H2D(m->dQKeys, KF->keys_un, KF->n); H2D(m->dQDesc, KF->desc, (size_t)KF->n * 32); H2D(m->dI[0], kf_mp, KF->n); }
    if (nmp > 0) H2D(m->dU8a, mp_bad, nmp);
    if (kf_fv->n_nodes > 0) { H2D(m->dNodesA, kf_fv->node_ids, kf_fv->n_nodes); H2D(m->dOffA, kf_fv->offsets, kf_fv->n_nodes + 1); }
    if (nqe > 0) H2D(m->dIdxA, kf_fv->indices, nqe);
    if (f_fv->n_nodes > 0) { H2D(m->dNodesB, f_fv->node_ids, f_fv->n_nodes); H2D(m->dOffB, f_fv->offsets, f_fv->n_nodes + 1); }
    if (nfe > 0) H2D(m->dFvIdx, f_fv->indices, nfe);
    m->gridPending = false;                                 // candidates come from the FeatureVectors: the spatial grid is not read
    FLUSH(m);
    if (kf_fv->n_nodes > 0)
        hipLaunchKernelGGL(k_queries_bow, dim3((std::max(nqe, 1) + 255) / 256), dim3(256), 0, nullptr, kf_fv->n_nodes, m->dNodesA, m->dOffA,
                           m->dIdxA, m->dI[0], m->dU8a, m->dQKeys, f_fv->n_nodes, m->dNodesB, m->dOffB, m->dQ);
    return run_search(m, MODE_BOW, nqe, fd, m->dQDesc, nullptr, nnratio, check_orientation, matches, nmatches_out);
}

extern "C" int rumi_search_by_bow_batch(RumiMatcher *m, int32_t K, const RumiFrameFeatures *KFs, const RumiFeatureVector *kf_fvs,
                                        const int32_t *const *kf_mp, const int32_t *nmp, const uint8_t *const *mp_bad, const RumiFrameFeatures *F,
                                        const RumiFeatureVector *f_fv, float nnratio, int32_t check_orientation, int32_t *matches,
                                        int32_t *nmatches_out) {
    if (!m || K < 1 || !KFs || !kf_fvs || !kf_mp || !nmp || !mp_bad || !F || !f_fv || !matches || !nmatches_out || F->n < 0) return RUMI_E_INVALID;
    HIP_TRY(hipSetDevice(m->device));
    const int nf = F->n, nnF = f_fv->n_nodes, nfe = nnF > 0 ? f_fv->offsets[nnF] : 0;
    for (int k = 0; k < K; k++) {
        nmatches_out[k] = 0;
        if (KFs[k].n < 0 || kf_fvs[k].n_nodes < 0 || nmp[k] < 0 || (KFs[k].n > 0 && !kf_mp[k])) return RUMI_E_INVALID;
    }
    for (size_t i = 0; i < (size_t)K * std::max(nf, 0); i++) matches[i] = -1;
    if (nf == 0 || nnF == 0) return RUMI_OK;
    // ---- one block: [frame arrays | key-frame table | key-frame arrays], every array on a 16-byte boundary ----
    size_t used = 0;
    auto take = [&](size_t bytes) { const size_t o = used; used += (bytes + 15) & ~(size_t)15; return o; };
    const size_t oFA = take((size_t)nf * 4), oFD = take((size_t)nf * 32), oFN = take((size_t)nnF * 4), oFO = take((size_t)(nnF + 1) * 4), oFI = take((size_t)nfe * 4);
    const size_t oT = take((size_t)K * sizeof(BowKF));
    std::vector<BowKF> tab(K);
    int maxNodes = 0;
    for (int k = 0; k < K; k++) {
        const int n = KFs[k].n, nn = kf_fvs[k].n_nodes, ne = nn > 0 ? kf_fvs[k].offsets[nn] : 0;
        BowKF &t = tab[k];
        t.n = n; t.nn = nn; t.pad = 0;
        t.angle = (int32_t)(take((size_t)n * 4) / 4); t.desc = (int32_t)(take((size_t)n * 32) / 4); t.mp = (int32_t)(take((size_t)n * 4) / 4);
        t.good = (int32_t)(take((size_t)n) / 4); t.nodes = (int32_t)(take((size_t)nn * 4) / 4); t.off = (int32_t)(take((size_t)(nn + 1) * 4) / 4);
        t.idx = (int32_t)(take((size_t)ne * 4) / 4);
        maxNodes = std::max(maxNodes, nn);
    }
    if (used > m->bowCap) {
        if (m->hBow) HIP_TRY(hipHostFree(m->hBow));
        if (m->dBow) HIP_TRY(hipFree(m->dBow));
        m->hBow = nullptr; m->dBow = nullptr; m->bowCap = 0;
        HIP_TRY(hipHostMalloc((void **)&m->hBow, used * 2, hipHostMallocDefault));
        HIP_TRY(hipMalloc((void **)&m->dBow, used * 2));
        m->bowCap = used * 2;
    }
    // outputs: [matches K nf | nmatch K | err 1 | hist K 32 | rotBin K nf bytes]; the first part comes back
    const size_t outInts = (size_t)K * nf + K + 1, outBytes = (outInts + (size_t)K * 32) * 4 + (size_t)K * nf;
    if (outBytes > m->bowOutCap) {
        if (m->hBowOut) HIP_TRY(hipHostFree(m->hBowOut));
        if (m->dBowOut) HIP_TRY(hipFree(m->dBowOut));
        m->hBowOut = nullptr; m->dBowOut = nullptr; m->bowOutCap = 0;
        HIP_TRY(hipHostMalloc((void **)&m->hBowOut, outBytes * 2, hipHostMallocDefault));
        HIP_TRY(hipMalloc((void **)&m->dBowOut, outBytes * 2));
        m->bowOutCap = outBytes * 2;
    }
    uint8_t *h = m->hBow;
    float *fa = reinterpret_cast<float *>(h + oFA);
    for (int i = 0; i < nf; i++) fa[i] = F->keys_un[i].angle;
    std::memcpy(h + oFD, F->desc, (size_t)nf * 32);
    std::memcpy(h + oFN, f_fv->node_ids, (size_t)nnF * 4);
    std::memcpy(h + oFO, f_fv->offsets, (size_t)(nnF + 1) * 4);
    if (nfe > 0) std::memcpy(h + oFI, f_fv->indices, (size_t)nfe * 4);
    std::memcpy(h + oT, tab.data(), (size_t)K * sizeof(BowKF));
    for (int k = 0; k < K; k++) {
        const BowKF &t = tab[k];
        const int n = t.n, nn = t.nn, ne = nn > 0 ? kf_fvs[k].offsets[nn] : 0;
        float *ka = reinterpret_cast<float *>(h + (size_t)t.angle * 4);
        uint8_t *good = h + (size_t)t.good * 4;
        for (int i = 0; i < n; i++) {
            ka[i] = KFs[k].keys_un[i].angle;
            const int mp = kf_mp[k][i];
            good[i] = mp >= 0 && mp < nmp[k] && !(mp_bad[k] && mp_bad[k][mp]);
        }
        if (n > 0) { std::memcpy(h + (size_t)t.desc * 4, KFs[k].desc, (size_t)n * 32); std::memcpy(h + (size_t)t.mp * 4, kf_mp[k], (size_t)n * 4); }
        if (nn > 0) { std::memcpy(h + (size_t)t.nodes * 4, kf_fvs[k].node_ids, (size_t)nn * 4); std::memcpy(h + (size_t)t.off * 4, kf_fvs[k].offsets, (size_t)(nn + 1) * 4); }
        if (ne > 0) std::memcpy(h + (size_t)t.idx * 4, kf_fvs[k].indices, (size_t)ne * 4);
    }
    HIP_TRY(hipMemcpyAsync(m->dBow, m->hBow, used, hipMemcpyHostToDevice, nullptr));
    int32_t *dOut = reinterpret_cast<int32_t *>(m->dBowOut);
    HIP_TRY(hipMemsetAsync(dOut, 0xFF, (size_t)K * nf * 4, nullptr));                              // matches = -1
    HIP_TRY(hipMemsetAsync(dOut + (size_t)K * nf, 0, ((size_t)K + 1 + (size_t)K * 32) * 4, nullptr));   // counts, error word, histograms
    BowBatch B;
    B.blk = reinterpret_cast<const uint32_t *>(m->dBow);
    B.K = K; B.nf = nf; B.nnF = nnF;
    B.fAngle = (int)(oFA / 4); B.fDesc = (int)(oFD / 4); B.fNodes = (int)(oFN / 4); B.fOff = (int)(oFO / 4); B.fIdx = (int)(oFI / 4); B.kfTable = (int)(oT / 4);
    B.matches = dOut; B.nmatch = dOut + (size_t)K * nf; B.err = B.nmatch + K; B.hist = B.err + 1;
    B.rotBin = reinterpret_cast<int8_t *>(B.hist + (size_t)K * 32);
    B.nnratio = nnratio; B.checkOri = check_orientation;
    if (maxNodes > 0) hipLaunchKernelGGL(k_bow_batch_match, dim3((maxNodes + 15) / 16, K), dim3(256), 0, nullptr, B);
    hipLaunchKernelGGL(k_bow_batch_finish, dim3(K), dim3(256), 0, nullptr, B);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(m->hBowOut, dOut, outInts * 4, hipMemcpyDeviceToHost));
    const int32_t *ho = reinterpret_cast<const int32_t *>(m->hBowOut);
    if (ho[(size_t)K * nf + K] & 1) {
        // a FeatureVector node of the frame holds more than 512 features (k_bow_batch_match keeps a node's "taken" flags in one 32-bit mask per
        // lane of a 16-lane group): shallow vocabularies or levelsup near L.  The results must still be those of K single searches, so run them.
        for (int k = 0; k < K; k++) {
            const int rc1 = rumi_search_by_bow(m, &KFs[k], &kf_fvs[k], kf_mp[k], nmp[k], mp_bad[k], F, f_fv, nnratio, check_orientation,
                                               matches + (size_t)k * nf, &nmatches_out[k]);
            if (rc1 != RUMI_OK) return rc1;
        }
        return RUMI_OK;
    }
    std::memcpy(matches, ho, (size_t)K * nf * 4);
    std::memcpy(nmatches_out, ho + (size_t)K * nf, (size_t)K * 4);
    return RUMI_OK;
}

extern "C" int rumi_search_by_bow_kf(RumiMatcher *m, const RumiFrameFeatures *KF1, const RumiFeatureVector *fv1, const int32_t *kf1_mp,
                                     const RumiFrameFeatures *KF2, const RumiFeatureVector *fv2, const int32_t *kf2_mp, int32_t nmp,
                                     const uint8_t *mp_bad, float nnratio, int32_t check_orientation, int32_t *matches12,
                                     int32_t *nmatches_out) {
    if (!m || !KF1 || !KF2 || !fv1 || !fv2 || !kf1_mp || !kf2_mp || !matches12 || !nmatches_out || nmp < 0) return RUMI_E_INVALID;
    const int nqe = fv1->n_nodes > 0 ? fv1->offsets[fv1->n_nodes] : 0, nfe = fv2->n_nodes > 0 ? fv2->offsets[fv2->n_nodes] : 0;
    if (KF1->n > m->maxQ || nqe > m->maxQ || nmp > m->maxQ || fv1->n_nodes > m->maxQ || nfe > m->maxFeat || fv2->n_nodes > m->maxFeat) {
        g_lastError = "SearchByBoW(KF,KF): sizes exceed the matcher's capacities";
        return RUMI_E_CAPACITY;
    }
    HIP_TRY(hipSetDevice(m->device));
    FrameDev fd;
    int rc = upload_frame(m, KF2, &fd);
    if (rc != RUMI_OK) return rc;
    // a KF2 feature is a candidate only if it holds a good map point (:732-736): everything else starts blocked
    std::vector<uint8_t> blocked(std::max(KF2->n, 1)), kf1bad(std::max(nmp, 1), 0);
    for (int f = 0; f < KF2->n; f++) blocked[f] = kf2_mp[f] < 0 || kf2_mp[f] >= nmp || mp_bad[kf2_mp[f]];
    if (KF2->n > 0) H2D(m->dU8b, blocked.data(), KF2->n);
    if (KF1->n > 0) { H2D(m->dQKeys, KF1->keys_un, KF1->n); H2D(m->dQDesc, KF1->desc, (size_t)KF1->n * 32); H2D(m->dI[0], kf1_mp, KF1->n); }
    if (nmp > 0) H2D(m->dU8a, mp_bad, nmp);
    if (fv1->n_nodes > 0) { H2D(m->dNodesA, fv1->node_ids, fv1->n_nodes); H2D(m->dOffA, fv1->offsets, fv1->n_nodes + 1); }
    if (nqe > 0) H2D(m->dIdxA, fv1->indices, nqe);
    if (fv2->n_nodes > 0) { H2D(m->dNodesB, fv2->node_ids, fv2->n_nodes); H2D(m->dOffB, fv2->offsets, fv2->n_nodes + 1); }
    if (nfe > 0) H2D(m->dFvIdx, fv2->indices, nfe);
    m->gridPending = false;
    FLUSH(m);
    if (fv1->n_nodes > 0)
        hipLaunchKernelGGL(k_queries_bow, dim3((std::max(nqe, 1) + 255) / 256), dim3(256), 0, nullptr, fv1->n_nodes, m->dNodesA, m->dOffA, m->dIdxA,
                           m->dI[0], m->dU8a, m->dQKeys, fv2->n_nodes, m->dNodesB, m->dOffB, m->dQ);
    std::vector<int32_t> assign(std::max(nqe, 1), -1);
    rc = run_search(m, MODE_BOW_KF, nqe, fd, m->dQDesc, nullptr, nnratio, check_orientation, nullptr, nmatches_out, m->dU8b, 0.f, 0, assign.data());
    if (rc != RUMI_OK) return rc;
    for (int i = 0; i < KF1->n; i++) matches12[i] = -1;
    for (int p = 0; p < nqe; p++) if (assign[p] >= 0) matches12[fv1->indices[p]] = assign[p];
    return RUMI_OK;
}

extern "C" int rumi_search_for_triangulation(RumiMatcher *m, const RumiFrameFeatures *KF1, const RumiFeatureVector *fv1, const int32_t *kf1_mp,
                                             const RumiFrameFeatures *KF2, const RumiFeatureVector *fv2, const int32_t *kf2_mp,
                                             const float *F12, const float *epipole2, int32_t only_stereo, int32_t coarse,
                                             int32_t check_orientation, int32_t *matches12, int32_t *nmatches_out) {
    if (!m || !KF1 || !KF2 || !fv1 || !fv2 || !F12 || !epipole2 || !nmatches_out) return RUMI_E_INVALID;
    if ((KF1->n > 0 && (!kf1_mp || !matches12)) || (KF2->n > 0 && !kf2_mp)) return RUMI_E_INVALID;
    const int nqe = fv1->n_nodes > 0 ? fv1->offsets[fv1->n_nodes] : 0, nfe = fv2->n_nodes > 0 ? fv2->offsets[fv2->n_nodes] : 0;
    if (KF1->n > m->maxQ || nqe > m->maxQ || fv1->n_nodes > m->maxQ || nfe > m->maxFeat || fv2->n_nodes > m->maxFeat) {
        g_lastError = "SearchForTriangulation: sizes exceed the matcher's capacities";
        return RUMI_E_CAPACITY;
    }
    for (int i = 0; i < KF1->n; i++) matches12[i] = -1;
    *nmatches_out = 0;
    // monocular key-frames have no stereo key-points (mvuRight < 0): bOnlyStereo skips every pair (:874-876)
    if (only_stereo || nqe == 0 || nfe == 0 || KF2->n == 0) return RUMI_OK;
    HIP_TRY(hipSetDevice(m->device));
    FrameDev fd;
    int rc = upload_frame(m, KF2, &fd);
    if (rc != RUMI_OK) return rc;
    float geom[11];
    std::memcpy(geom, F12, 36); std::memcpy(geom + 9, epipole2, 8);
    H2D(m->dPose, geom, 11);
    H2D(m->dFeatMp, kf2_mp, KF2->n);
    H2D(m->dQKeys, KF1->keys_un, KF1->n); H2D(m->dQDesc, KF1->desc, (size_t)KF1->n * 32); H2D(m->dI[0], kf1_mp, KF1->n);
    H2D(m->dNodesA, fv1->node_ids, fv1->n_nodes); H2D(m->dOffA, fv1->offsets, fv1->n_nodes + 1); H2D(m->dIdxA, fv1->indices, nqe);
    H2D(m->dNodesB, fv2->node_ids, fv2->n_nodes); H2D(m->dOffB, fv2->offsets, fv2->n_nodes + 1); H2D(m->dFvIdx, fv2->indices, nfe);
    TriArgs A{fv1->n_nodes, fv2->n_nodes, m->dNodesA, m->dOffA, m->dIdxA, m->dNodesB, m->dOffB, m->dFvIdx, m->dQKeys, m->dKeys,
              m->dQDesc, m->dDesc, m->dI[0], m->dFeatMp, m->dScale, m->dPose, coarse, m->dAssign};
    FLUSH(m);
    hipLaunchKernelGGL(k_tri_match, dim3((fv1->n_nodes + 63) / 64), dim3(64), 0, nullptr, A);
    hipLaunchKernelGGL(k_tri_filter, dim3(1), dim3(256), 0, nullptr, nqe, m->dIdxA, m->dQKeys, m->dKeys, m->dAssign, check_orientation, m->dNmatches);
    HIP_TRY(hipGetLastError());
    rc = fetch_results(m, 0, nqe, true);
    if (rc != RUMI_OK) return rc;
    *nmatches_out = m->hOut[0];
    const int32_t *assign = m->hOut + 4 + m->maxFeat;
    for (int p = 0; p < nqe; p++) if (assign[p] >= 0) matches12[fv1->indices[p]] = assign[p];
    return RUMI_OK;
}

extern "C" int rumi_search_by_projection_sim3(RumiMatcher *m, const RumiFrameFeatures *KF, float log_scale_factor, const float *Tcw7,
                                              const float *Ow3, const float *K4, int32_t nmp, const uint8_t *skip, const float *mp_pos,
                                              const float *mp_normal, const float *mp_min_dist, const float *mp_max_dist,
                                              const uint8_t *mp_desc, int32_t th, float ratio_hamming, int32_t explicit_invz,
                                              int32_t *matched, int32_t *nmatches_out) {
    if (!m || !KF || !Tcw7 || !Ow3 || !K4 || !matched || !nmatches_out || nmp < 0) return RUMI_E_INVALID;
    if (nmp > m->maxQ) { g_lastError = "more candidate points than max_queries"; return RUMI_E_CAPACITY; }
    HIP_TRY(hipSetDevice(m->device));
    FrameDev fd;
    int rc = upload_frame(m, KF, &fd);
    if (rc != RUMI_OK) return rc;
    std::vector<uint8_t> blocked(std::max(KF->n, 1));
    for (int f = 0; f < KF->n; f++) blocked[f] = matched[f] != -1;                 // vpMatched[idx] != NULL (:442)
    if (KF->n > 0) { H2D(m->dU8b, blocked.data(), KF->n); H2D(m->dFeatMp, matched, KF->n); }
    float pose[14];
    std::memcpy(pose, Tcw7, 28); std::memcpy(pose + 7, K4, 16); std::memcpy(pose + 11, Ow3, 12);
    H2D(m->dPose, pose, 14);
    if (nmp > 0) {
        H2D(m->dU8a, skip, nmp); H2D(m->dF[0], mp_pos, (size_t)nmp * 3); H2D(m->dF[1], mp_normal, (size_t)nmp * 3);
        H2D(m->dF[2], mp_min_dist, nmp); H2D(m->dF[3], mp_max_dist, nmp); H2D(m->dQDesc, mp_desc, (size_t)nmp * 32);
        FLUSH(m);
        hipLaunchKernelGGL(k_queries_sim3, dim3((nmp + 255) / 256), dim3(256), 0, nullptr, nmp, m->dU8a, m->dF[0], m->dF[1], m->dF[2], m->dF[3],
                           m->dPose, m->dScale, KF->nlevels, log_scale_factor, (float)th, explicit_invz, 1, 0, fd.minX, fd.minY, fd.maxX, fd.maxY, m->dQ);
    }
    return run_search(m, MODE_SIM3, nmp, fd, m->dQDesc, nullptr, 0.f, 0, matched, nmatches_out, m->dU8b, (float)RUMI_TH_LOW * ratio_hamming, 0);
}

extern "C" int rumi_fuse_candidates(RumiMatcher *m, const RumiFrameFeatures *KF, float log_scale_factor, const float *Tcw7, const float *Ow3,
                                    const float *K4, int32_t nmp, const uint8_t *skip, const float *mp_pos, const float *mp_normal,
                                    const float *mp_min_dist, const float *mp_max_dist, const uint8_t *mp_desc, float th,
                                    int32_t check_reprojection, int32_t *best_idx) {
    if (!m || !KF || !Tcw7 || !Ow3 || !K4 || nmp < 0 || (nmp > 0 && !best_idx)) return RUMI_E_INVALID;
    if (nmp > m->maxQ) { g_lastError = "more candidate points than max_queries"; return RUMI_E_CAPACITY; }
    if (nmp == 0) return RUMI_OK;
    HIP_TRY(hipSetDevice(m->device));
    FrameDev fd;
    int rc = upload_frame(m, KF, &fd);
    if (rc != RUMI_OK) return rc;
    HIP_TRY(hipMemsetAsync(m->dU8b, 0, std::max(KF->n, 1), nullptr));              // nothing is blocked: the points do not compete
    float pose[14];
    std::memcpy(pose, Tcw7, 28); std::memcpy(pose + 7, K4, 16); std::memcpy(pose + 11, Ow3, 12);
    H2D(m->dPose, pose, 14);
    H2D(m->dU8a, skip, nmp); H2D(m->dF[0], mp_pos, (size_t)nmp * 3); H2D(m->dF[1], mp_normal, (size_t)nmp * 3);
    H2D(m->dF[2], mp_min_dist, nmp); H2D(m->dF[3], mp_max_dist, nmp); H2D(m->dQDesc, mp_desc, (size_t)nmp * 32);
    FLUSH(m);
    hipLaunchKernelGGL(k_queries_sim3, dim3((nmp + 255) / 256), dim3(256), 0, nullptr, nmp, m->dU8a, m->dF[0], m->dF[1], m->dF[2], m->dF[3],
                       m->dPose, m->dScale, KF->nlevels, log_scale_factor, th, 0, 0, check_reprojection, fd.minX, fd.minY, fd.maxX, fd.maxY, m->dQ);
    int32_t n = 0;
    return run_search(m, MODE_FUSE, nmp, fd, m->dQDesc, nullptr, 0.f, 0, nullptr, &n, m->dU8b, 0.f, RUMI_TH_LOW, best_idx);
}

static int sim3_direction(RumiMatcher *m, const RumiFrameFeatures *KF, float logSf, const float *K4, int n, const uint8_t *skip, const float *pc,
                          const float *mn, const float *mx, const uint8_t *desc, float th, int32_t *best) {
    for (int i = 0; i < n; i++) best[i] = -1;
    if (n == 0) return RUMI_OK;
    FrameDev fd;
    int rc = upload_frame(m, KF, &fd);
    if (rc != RUMI_OK) return rc;
    HIP_TRY(hipMemsetAsync(m->dU8b, 0, std::max(KF->n, 1), nullptr));
    H2D(m->dPose, K4, 4);
    H2D(m->dU8a, skip, n); H2D(m->dF[0], pc, (size_t)n * 3); H2D(m->dF[2], mn, n); H2D(m->dF[3], mx, n); H2D(m->dQDesc, desc, (size_t)n * 32);
    FLUSH(m);
    hipLaunchKernelGGL(k_queries_campoints, dim3((n + 255) / 256), dim3(256), 0, nullptr, n, m->dU8a, m->dF[0], m->dF[2], m->dF[3], m->dPose, m->dScale,
                       KF->nlevels, logSf, th, fd.minX, fd.minY, fd.maxX, fd.maxY, m->dQ);
    int32_t cnt = 0;
    return run_search(m, MODE_FUSE, n, fd, m->dQDesc, nullptr, 0.f, 0, nullptr, &cnt, m->dU8b, 0.f, RUMI_TH_HIGH, best);
}

extern "C" int rumi_search_by_sim3(RumiMatcher *m, const RumiFrameFeatures *KF1, const RumiFrameFeatures *KF2, const float *K4,
                                   float log_scale_factor, const uint8_t *skip1, const float *pc1_in2, const float *min_dist1,
                                   const float *max_dist1, const uint8_t *desc1, const uint8_t *skip2, const float *pc2_in1,
                                   const float *min_dist2, const float *max_dist2, const uint8_t *desc2, float th, int32_t *match12,
                                   int32_t *nfound_out) {
    if (!m || !KF1 || !KF2 || !K4 || !nfound_out) return RUMI_E_INVALID;
    const int n1 = KF1->n, n2 = KF2->n;
    if (n1 < 0 || n2 < 0 || (n1 > 0 && (!skip1 || !pc1_in2 || !min_dist1 || !max_dist1 || !desc1 || !match12)) ||
        (n2 > 0 && (!skip2 || !pc2_in1 || !min_dist2 || !max_dist2 || !desc2)))
        return RUMI_E_INVALID;
    if (n1 > m->maxQ || n2 > m->maxQ) { g_lastError = "SearchBySim3: key-frame larger than max_queries"; return RUMI_E_CAPACITY; }
    HIP_TRY(hipSetDevice(m->device));
    std::vector<int32_t> vnMatch1(std::max(n1, 1)), vnMatch2(std::max(n2, 1));
    int rc = sim3_direction(m, KF2, log_scale_factor, K4, n1, skip1, pc1_in2, min_dist1, max_dist1, desc1, th, vnMatch1.data());
    if (rc != RUMI_OK) return rc;
    rc = sim3_direction(m, KF1, log_scale_factor, K4, n2, skip2, pc2_in1, min_dist2, max_dist2, desc2, th, vnMatch2.data());
    if (rc != RUMI_OK) return rc;
    int nFound = 0;                                                                 // check agreement, :1480-1493
    for (int i1 = 0; i1 < n1; i1++) {
        match12[i1] = -1;
        const int idx2 = vnMatch1[i1];
        if (idx2 >= 0 && vnMatch2[idx2] == i1) { match12[i1] = idx2; nFound++; }
    }
    *nfound_out = nFound;
    return RUMI_OK;
}

extern "C" int rumi_search_by_projection_reloc(RumiMatcher *m, const RumiFrameFeatures *Cur, float log_scale_factor, const float *Tcw7,
                                               const float *Ow3, const float *K4, const RumiKeyPoint *kf_keys, int32_t nkf,
                                               const int32_t *kf_mp, int32_t nmp, const uint8_t *skip, const float *mp_pos,
                                               const float *mp_min_dist, const float *mp_max_dist, const uint8_t *mp_desc, float th,
                                               int32_t orb_dist, int32_t check_orientation, int32_t *cur_mp, int32_t *nmatches_out) {
    if (!m || !Cur || !Tcw7 || !Ow3 || !K4 || !cur_mp || !nmatches_out || nkf < 0 || nmp < 0) return RUMI_E_INVALID;
    if (nkf > m->maxQ || nmp > m->maxQ) { g_lastError = "more key-frame features / map points than max_queries"; return RUMI_E_CAPACITY; }
    HIP_TRY(hipSetDevice(m->device));
    FrameDev fd;
    int rc = upload_frame(m, Cur, &fd);
    if (rc != RUMI_OK) return rc;
    std::vector<uint8_t> blocked(std::max(Cur->n, 1));
    for (int f = 0; f < Cur->n; f++) blocked[f] = cur_mp[f] >= 0;                  // CurrentFrame.mvpMapPoints[i2] != NULL (:1746)
    if (Cur->n > 0) { H2D(m->dU8b, blocked.data(), Cur->n); H2D(m->dFeatMp, cur_mp, Cur->n); }
    float pose[14];
    std::memcpy(pose, Tcw7, 28); std::memcpy(pose + 7, K4, 16); std::memcpy(pose + 11, Ow3, 12);
    H2D(m->dPose, pose, 14);
    if (nmp > 0) {
        H2D(m->dU8a, skip, nmp); H2D(m->dF[0], mp_pos, (size_t)nmp * 3); H2D(m->dF[2], mp_min_dist, nmp); H2D(m->dF[3], mp_max_dist, nmp);
        H2D(m->dQDesc, mp_desc, (size_t)nmp * 32);
    }
    if (nkf > 0) {
        H2D(m->dQKeys, kf_keys, nkf); H2D(m->dI[0], kf_mp, nkf);
        FLUSH(m);
        hipLaunchKernelGGL(k_queries_reloc, dim3((nkf + 255) / 256), dim3(256), 0, nullptr, nkf, m->dQKeys, m->dI[0], m->dU8a, m->dF[0], m->dF[2],
                           m->dF[3], m->dPose, m->dScale, Cur->nlevels, log_scale_factor, th, fd.minX, fd.minY, fd.maxX, fd.maxY, m->dQ);
    }
    return run_search(m, MODE_RELOC, nkf, fd, m->dQDesc, nullptr, 0.f, check_orientation, cur_mp, nmatches_out, m->dU8b, 0.f, orb_dist);
}

extern "C" int rumi_search_for_initialization(RumiMatcher *m, const RumiFrameFeatures *F1, const RumiFrameFeatures *F2,
                                              float *prev_matched, int32_t window_size, float nnratio, int32_t check_orientation,
                                              int32_t *matches12, int32_t *nmatches_out) {
    if (!m || !F1 || !F2 || !nmatches_out || F1->n < 0) return RUMI_E_INVALID;
    if (F1->n > m->maxQ) { g_lastError = "more F1 key-points than max_queries"; return RUMI_E_CAPACITY; }
    if (F1->n > 0 && (!prev_matched || !matches12 || !F1->keys_un || !F1->desc)) return RUMI_E_INVALID;
    HIP_TRY(hipSetDevice(m->device));
    FrameDev fd;
    int rc = upload_frame(m, F2, &fd);
    if (rc != RUMI_OK) return rc;
    const int n1 = F1->n;
    *nmatches_out = 0;
    if (n1 == 0) return RUMI_OK;
    H2D(m->dQKeys, F1->keys_un, n1); H2D(m->dQDesc, F1->desc, (size_t)n1 * 32); H2D(m->dF[0], prev_matched, (size_t)n1 * 2);
    FLUSH(m);
    hipLaunchKernelGGL(k_queries_init, dim3((n1 + 255) / 256), dim3(256), 0, nullptr, n1, m->dQKeys, m->dF[0], (float)window_size, m->dQ);
    for (int attempt = 0; attempt < 2; attempt++) {
        rc = build_lists(m, MODE_INIT, n1, fd, m->dQDesc, attempt > 0, false);
        if (rc != RUMI_OK) return rc;
        InitArgs A{n1, fd.n, m->dQ, m->dCounts, m->dOffsets, m->dLists, fd.keys, m->dAssign, m->dF[0], m->dNmatches, nnratio, check_orientation,
                   m->dOverflow};
        hipLaunchKernelGGL(k_resolve_init, dim3(1), dim3(64), (size_t)std::max(fd.n, 1) * 2 * sizeof(int32_t), nullptr, A);
        HIP_TRY(hipGetLastError());
        rc = fetch_results(m, 0, n1, true);
        if (rc != RUMI_OK) return rc;
        if (m->hOut[1] == 0) break;
        if (attempt == 1) { g_lastError = "candidate list arena overflow after growing"; return RUMI_E_CAPACITY; }
        rc = grow_lists(m, (size_t)m->hOut[1]);
        if (rc != RUMI_OK) return rc;
    }
    *nmatches_out = m->hOut[0];
    std::memcpy(matches12, m->hOut + 4 + m->maxFeat, (size_t)n1 * sizeof(int32_t));
    HIP_TRY(hipMemcpy(prev_matched, m->dF[0], (size_t)n1 * 2 * sizeof(float), hipMemcpyDeviceToHost));
    return RUMI_OK;
}

extern "C" int rumi_frame_is_in_frustum(RumiMatcher *m, const float *Rcw9, const float *tcw3, const float *Ow3, const float *K4,
                                        float min_x, float min_y, float max_x, float max_y, float log_scale_factor, int32_t nlevels,
                                        float viewing_cos_limit, int32_t nmp, const float *mp_pos, const float *mp_normal,
                                        const float *mp_min_dist, const float *mp_max_dist, uint8_t *track_in_view, float *proj_x,
                                        float *proj_y, int32_t *scale_level, float *view_cos, float *track_depth) {
    if (!m || !Rcw9 || !tcw3 || !Ow3 || !K4 || nmp < 0) return RUMI_E_INVALID;
    if (nmp > m->maxQ) { g_lastError = "more map points than max_queries"; return RUMI_E_CAPACITY; }
    if (nmp == 0) return RUMI_OK;
    if (!mp_pos || !mp_normal || !mp_min_dist || !mp_max_dist || !track_in_view || !proj_x || !proj_y || !scale_level || !view_cos || !track_depth)
        return RUMI_E_INVALID;
    HIP_TRY(hipSetDevice(m->device));
    reset_uploads(m);
    float pose[19];
    std::memcpy(pose, Rcw9, 36); std::memcpy(pose + 9, tcw3, 12); std::memcpy(pose + 12, Ow3, 12); std::memcpy(pose + 15, K4, 16);
    H2D(m->dPose, pose, 19);
    H2D(m->dF[0], mp_pos, (size_t)nmp * 3); H2D(m->dF[1], mp_normal, (size_t)nmp * 3); H2D(m->dF[2], mp_min_dist, nmp); H2D(m->dF[3], mp_max_dist, nmp);
    // outputs are packed into the upload mirror (its contents have been scattered by then) and come back with one copy
    const size_t n16 = ((size_t)nmp + 15) & ~(size_t)15;
    if (n16 * 21 > m->stageCap) { g_lastError = "isInFrustum: result block exceeds the staging block"; return RUMI_E_CAPACITY; }
    uint8_t *dIn = m->dStage;
    float *dX = reinterpret_cast<float *>(m->dStage + n16), *dY = dX + n16, *dC = dY + n16, *dD = dC + n16;
    int32_t *dL = reinterpret_cast<int32_t *>(dD + n16);
    FLUSH(m);
    hipLaunchKernelGGL(k_is_in_frustum, dim3((nmp + 255) / 256), dim3(256), 0, nullptr, nmp, m->dPose, min_x, min_y, max_x, max_y, log_scale_factor,
                       nlevels, viewing_cos_limit, m->dF[0], m->dF[1], m->dF[2], m->dF[3], dIn, dX, dY, dL, dC, dD);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(m->hStage, m->dStage, n16 * 21, hipMemcpyDeviceToHost));
    const uint8_t *h = m->hStage;
    std::memcpy(track_in_view, h, (size_t)nmp);
    std::memcpy(proj_x, h + n16, (size_t)nmp * 4); std::memcpy(proj_y, h + n16 * 5, (size_t)nmp * 4);
    std::memcpy(view_cos, h + n16 * 9, (size_t)nmp * 4); std::memcpy(track_depth, h + n16 * 13, (size_t)nmp * 4);
    std::memcpy(scale_level, h + n16 * 17, (size_t)nmp * 4);
    return RUMI_OK;
}

extern "C" int rumi_search_local_points(RumiMatcher *m, const RumiFrameFeatures *F, const float *Rcw9, const float *tcw3, const float *Ow3,
                                        const float *K4, float log_scale_factor, int32_t nlevels, float viewing_cos_limit, int32_t nmp,
                                        const uint8_t *skip, const float *mp_pos, const float *mp_normal, const float *mp_min_dist,
                                        const float *mp_max_dist, const uint8_t *mp_desc, const int32_t *mp_obs, float th, int32_t far_points,
                                        float th_far_points, float nnratio, uint8_t *track_in_view, float *proj_x, float *proj_y,
                                        int32_t *scale_level, float *view_cos, float *track_depth, int32_t *n_to_match_out, int32_t *frame_mp,
                                        int32_t *nmatches_out) {
    if (!m || !F || !Rcw9 || !tcw3 || !Ow3 || !K4 || !nmatches_out || !n_to_match_out || !frame_mp || nmp < 0) return RUMI_E_INVALID;
    *nmatches_out = 0; *n_to_match_out = 0;
    if (nmp > m->maxQ) { g_lastError = "more map points than max_queries"; return RUMI_E_CAPACITY; }
    if (nmp == 0) return RUMI_OK;                          // nToMatch == 0: the reference does not search (Tracking.cc:3032)
    if (!skip || !mp_pos || !mp_normal || !mp_min_dist || !mp_max_dist || !mp_desc || !mp_obs || !track_in_view || !proj_x || !proj_y || !scale_level ||
        !view_cos || !track_depth)
        return RUMI_E_INVALID;
    HIP_TRY(hipSetDevice(m->device));
    FrameDev fd;
    int rc = upload_frame(m, F, &fd);
    if (rc != RUMI_OK) return rc;
    if (F->n > 0) H2D(m->dFeatMp, frame_mp, F->n);
    float pose[19];
    std::memcpy(pose, Rcw9, 36); std::memcpy(pose + 9, tcw3, 12); std::memcpy(pose + 12, Ow3, 12); std::memcpy(pose + 15, K4, 16);
    H2D(m->dPose, pose, 19);
    H2D(m->dF[0], mp_pos, (size_t)nmp * 3); H2D(m->dF[1], mp_normal, (size_t)nmp * 3); H2D(m->dF[2], mp_min_dist, nmp); H2D(m->dF[3], mp_max_dist, nmp);
    H2D(m->dU8b, skip, nmp); H2D(m->dI[1], mp_obs, nmp); H2D(m->dQDesc, mp_desc, (size_t)nmp * 32);
    // the frustum test writes the six per-point fields into the (by then scattered) upload mirror; the query kernel reads them there and the
    // same block travels back to the host for the facade's write-back: no host round trip between isInFrustum and SearchByProjection
    const size_t n16 = ((size_t)nmp + 15) & ~(size_t)15;
    if (n16 * 21 > m->stageCap) { g_lastError = "SearchLocalPoints: result block exceeds the staging block"; return RUMI_E_CAPACITY; }
    uint8_t *dIn = m->dStage;
    float *dX = reinterpret_cast<float *>(m->dStage + n16), *dY = dX + n16, *dC = dY + n16, *dD = dC + n16;
    int32_t *dL = reinterpret_cast<int32_t *>(dD + n16);
    FLUSH(m);
    hipLaunchKernelGGL(k_is_in_frustum, dim3((nmp + 255) / 256), dim3(256), 0, nullptr, nmp, m->dPose, fd.minX, fd.minY, fd.maxX, fd.maxY, log_scale_factor,
                       nlevels, viewing_cos_limit, m->dF[0], m->dF[1], m->dF[2], m->dF[3], dIn, dX, dY, dL, dC, dD, m->dU8b);
    HIP_TRY(hipMemcpyAsync(m->hStage, m->dStage, n16 * 21, hipMemcpyDeviceToHost, nullptr));
    // is_bad of SearchByProjection = skip: a skipped point is never in view, so the flag is only read for points that are not bad
    hipLaunchKernelGGL(k_queries_mappoints, dim3((nmp + 255) / 256), dim3(256), 0, nullptr, nmp, dIn, dX, dY, dL, dC, dD, m->dU8b, m->dI[1], m->dScale, th,
                       far_points, th_far_points, m->dQ);
    rc = run_search(m, MODE_MAPPOINTS, nmp, fd, m->dQDesc, m->dI[1], nnratio, 0, frame_mp, nmatches_out);
    if (rc != RUMI_OK) return rc;
    const uint8_t *h = m->hStage;                           // complete: run_search synchronised the stream
    std::memcpy(track_in_view, h, (size_t)nmp);
    std::memcpy(proj_x, h + n16, (size_t)nmp * 4); std::memcpy(proj_y, h + n16 * 5, (size_t)nmp * 4);
    std::memcpy(view_cos, h + n16 * 9, (size_t)nmp * 4); std::memcpy(track_depth, h + n16 * 13, (size_t)nmp * 4);
    std::memcpy(scale_level, h + n16 * 17, (size_t)nmp * 4);
    int nTo = 0;
    for (int i = 0; i < nmp; i++) nTo += track_in_view[i];
    *n_to_match_out = nTo;
    if (nTo == 0) *nmatches_out = 0;                       // (nothing in view: no query was live, the search found nothing)
    return RUMI_OK;
}

extern "C" int rumi_match_bruteforce_batch_device_strided(const void *d_query, const void *d_nq, const void *d_train, const void *d_nt,
                                                          int32_t count_stride, int64_t query_stride, int64_t train_stride, int32_t cap, int32_t nbatch,
                                                          void *d_best_idx, void *d_best_dist, void *d_second_dist, void *hip_stream) {
    if (!d_query || !d_nq || !d_train || !d_nt || !d_best_idx || !d_best_dist || !d_second_dist || cap < 1 || cap > 65535 || nbatch < 1 || count_stride < 1 ||
        query_stride < 32ll * cap || train_stride < 32ll * cap || (query_stride & 3) || (train_stride & 3) ||
        (reinterpret_cast<uintptr_t>(d_query) & 3) || (reinterpret_cast<uintptr_t>(d_train) & 3))
        return RUMI_E_INVALID;                                 // the kernel packs the train index into 16 bits next to the distance
    hipLaunchKernelGGL(k_bruteforce, dim3((cap + 255) / 256, nbatch), dim3(256), 0, (hipStream_t)hip_stream, (const uint8_t *)d_query,
                       (const int32_t *)d_nq, (const uint8_t *)d_train, (const int32_t *)d_nt, count_stride, (long long)query_stride, (long long)train_stride, cap,
                       (int32_t *)d_best_idx, (int32_t *)d_best_dist, (int32_t *)d_second_dist, 0);
    HIP_TRY(hipGetLastError());
    return RUMI_OK;
}

extern "C" int rumi_match_bruteforce_ring_device(const void *d_desc, const void *d_n, int32_t count_stride, int64_t frame_stride, int32_t cap, int32_t nframes,
                                                 void *d_best_idx, void *d_best_dist, void *d_second_dist, void *hip_stream) {
    if (!d_desc || !d_n || !d_best_idx || !d_best_dist || !d_second_dist || cap < 1 || cap > 65535 || nframes < 1 || count_stride < 1 ||
        frame_stride < 32ll * cap || (frame_stride & 3) || (reinterpret_cast<uintptr_t>(d_desc) & 3))
        return RUMI_E_INVALID;
    hipLaunchKernelGGL(k_bruteforce, dim3((cap + 255) / 256, nframes), dim3(256), 0, (hipStream_t)hip_stream, (const uint8_t *)d_desc, (const int32_t *)d_n,
                       (const uint8_t *)d_desc, (const int32_t *)d_n, count_stride, (long long)frame_stride, (long long)frame_stride, cap, (int32_t *)d_best_idx,
                       (int32_t *)d_best_dist, (int32_t *)d_second_dist, nframes);
    HIP_TRY(hipGetLastError());
    return RUMI_OK;
}

extern "C" int rumi_match_bruteforce_batch_device(const void *d_query, const void *d_nq, const void *d_train, const void *d_nt,
                                                  int32_t count_stride, int32_t cap, int32_t nbatch, void *d_best_idx,
                                                  void *d_best_dist, void *d_second_dist, void *hip_stream) {
    return rumi_match_bruteforce_batch_device_strided(d_query, d_nq, d_train, d_nt, count_stride, 32ll * cap, 32ll * cap, cap, nbatch, d_best_idx, d_best_dist,
                                                      d_second_dist, hip_stream);
}

// ==================================================================================================================
// One device-resident Tracking step (include/rumi_track.h): the extractor's record, the matcher's grid and map-point vector (mvpMapPoints =
// dFeatMp) and the pose optimiser's correspondence arrays never leave HBM between the five stages.  A dispatch costs about 4.5 us on the
// device whatever it does, so the step is built from as few as the data flow allows: no device-to-device copies (the matcher reads the
// extractor's record in place, results are produced inside the block that travels back), fills and bookkeeping folded into neighbouring kernels.
// ==================================================================================================================
namespace rumi {

constexpr int kTrackLdsEdges = rumi::kPoseLdsEdges;     // (rumi_internal.h: one number for both files)
struct TrackBlock {                  // the result block's header, device and pinned host alike (arrays follow at byte offsets of RumiTracker)
    float Tout[14];                  // pose after the motion model | after the local map
    float pose19[20];                // Rcw9 tcw3 Ow3 K4 of the first (Frame::UpdatePoseMatrices)
    int32_t nGood[2];                // PoseOptimization return values
    int32_t counters[2];             // nmatchesMap, mnMatchesInliers
    int32_t start[2];                // correspondences of the optimisation in flight: {0, count}
    int32_t spec[4];                 // speculative step: result header (matches, overflow word) of the motion search | of the local search
};

// fill(mvpMapPoints, NULL), cleared flags / counters, both poses = the prediction, result header of the search cleared
__global__ void k_track_init(int n, int nmp, int full, int32_t *featMp, int32_t *searchHeader, uint8_t *seen, uint8_t *outF, int32_t *mpOut, const float *Tpred,
                             TrackBlock *blk) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { featMp[i] = -1; if (full) { outF[i] = 0; mpOut[i] = -1; } }
    if (full && i < nmp) seen[i] = 0;
    if (i < 4) searchHeader[i] = 0;
    if (full && i == 0) {
        for (int k = 0; k < 7; k++) { blk->Tout[k] = Tpred[k]; blk->Tout[7 + k] = Tpred[k]; }
        for (int k = 0; k < 20; k++) blk->pose19[k] = 0.f;
        blk->nGood[0] = blk->nGood[1] = 0; blk->counters[0] = blk->counters[1] = 0; blk->start[0] = blk->start[1] = 0;
    }
}

// (gather_correspondences as a launch of its own: the paths that do not end a search with it)
__global__ __launch_bounds__(1024) void k_track_gather(int n, const RumiKeyPoint *__restrict__ keys, const int32_t *__restrict__ featMp,
                                                       const float *__restrict__ mpPos, const float *__restrict__ invSigma2, float *Xw, float *obs,
                                                       float *w, int32_t *idx, int32_t *start, int32_t *snapshot = nullptr) {
    __shared__ int sWave[16], sBase;
    gather_correspondences(n, keys, featMp, mpPos, invSigma2, Xw, obs, w, idx, start, snapshot, sWave, &sBase);
}

// Frame::UpdatePoseMatrices (Frame.cc:522-528) in Sophus' / Eigen's float arithmetic: Rcw = q.toRotationMatrix(), tcw, Ow = conj(q) * (-tcw)
// (quaternion _transformVector), as [Rcw9 | tcw3 | Ow3 | K4] for the frustum test.
__device__ __forceinline__ void pose_matrices19(const float *Tcw7, const float *K4, float *pose19) {
    const float x = Tcw7[0], y = Tcw7[1], z = Tcw7[2], w = Tcw7[3];
    const float tx = 2.f * x, ty = 2.f * y, tz = 2.f * z;
    const float twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    pose19[0] = 1.f - (tyy + tzz); pose19[1] = txy - twz; pose19[2] = txz + twy;
    pose19[3] = txy + twz; pose19[4] = 1.f - (txx + tzz); pose19[5] = tyz - twx;
    pose19[6] = txz - twy; pose19[7] = tyz + twx; pose19[8] = 1.f - (txx + tyy);
    const float t0 = Tcw7[4], t1 = Tcw7[5], t2 = Tcw7[6];
    pose19[9] = t0; pose19[10] = t1; pose19[11] = t2;
    const float qx = -x, qy = -y, qz = -z, v0 = t0 * -1.f, v1 = t1 * -1.f, v2 = t2 * -1.f;
    float u0 = qy * v2 - qz * v1, u1 = qz * v0 - qx * v2, u2 = qx * v1 - qy * v0;
    u0 += u0; u1 += u1; u2 += u2;
    const float c0 = qy * u2 - qz * u1, c1 = qz * u0 - qx * u2, c2 = qx * u1 - qy * u0;
    pose19[12] = (v0 + w * u0) + c0; pose19[13] = (v1 + w * u1) + c1; pose19[14] = (v2 + w * u2) + c2;
    pose19[15] = K4[0]; pose19[16] = K4[1]; pose19[17] = K4[2]; pose19[18] = K4[3];
}

// "Discard outliers" of TrackWithMotionModel / TrackReferenceKeyFrame alone (Tracking.cc:2489-2508, 2349-2369): the outliers of the optimisation
// leave the frame, the others count towards nmatchesMap when their point has observations.  (The step-wise entries: SearchLocalPoints' own
// loops belong to rumi_track_local.)
__global__ void k_track_discard(const int32_t *idx, const uint8_t *outlierC, int32_t *featMp, const int32_t *mpObs, TrackBlock *blk,
                                const int32_t *searchHeader = nullptr) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && searchHeader) { blk->spec[0] = searchHeader[0]; blk->spec[1] = searchHeader[1]; }
    if (c >= blk->start[1]) return;
    const int i = idx[c], mp = featMp[i];
    if (outlierC[c]) { featMp[i] = -1; return; }
    if (mpObs[mp] > 0) atomicAdd(&blk->counters[0], 1);
}

// rumi_track_local: the pose the stage starts from and its UpdatePoseMatrices, cleared outputs and counters (the frame's map-point vector and
// the seen flags arrive with the stage's upload)
__global__ void k_track_local_init(int n, const float *Tcw7, const float *K4, uint8_t *outF, int32_t *mpOut, int32_t *searchHeader, TrackBlock *blk) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { outF[i] = 0; mpOut[i] = -1; }
    if (i < 4) searchHeader[i] = 0;
    if (i == 0) {
        for (int k = 0; k < 7; k++) { blk->Tout[k] = Tcw7[k]; blk->Tout[7 + k] = Tcw7[k]; }
        pose_matrices19(Tcw7, K4, blk->pose19);
        blk->nGood[0] = blk->nGood[1] = 0; blk->counters[0] = blk->counters[1] = 0; blk->start[0] = blk->start[1] = 0;
    }
}

// The frame's DBoW2::FeatureVector on the device (TemplatedVocabulary.h:1147-1190, FeatureVector.cpp:31-45): the features with a positive word
// weight grouped by their node id, nodes ascending, feature indices ascending inside a node, in the CSR form the BoW search reads.
// ONE workgroup: 64-bit keys node << 32 | feature in LDS, bitonic sort, then the group boundaries by an ordered compaction.
constexpr int kFvThreads = 1024;
__global__ __launch_bounds__(kFvThreads) void k_fv_build(int n, int npad, const uint32_t *__restrict__ node, const double *__restrict__ weight,
                                                         uint32_t *fvNodes, int32_t *fvOff, uint32_t *fvIdx, int32_t *nnOut) {
    extern __shared__ unsigned long long fvKey[];          // npad keys (a power of two >= n)
    __shared__ int sCnt[kFvThreads / 64], sBase, sValid;
    const int tid = threadIdx.x;
    for (int i = tid; i < npad; i += kFvThreads)
        fvKey[i] = (i < n && weight[i] > 0.0) ? (((unsigned long long)node[i] << 32) | (unsigned)i) : ~0ull;       // stopped words and padding sort last
    __syncthreads();
    for (int k = 2; k <= npad; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < npad; i += kFvThreads) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long a = fvKey[i], b = fvKey[l];
                    if ((a > b) == ((i & k) == 0)) { fvKey[i] = b; fvKey[l] = a; }
                }
            }
            __syncthreads();
        }
    if (tid == 0) { sBase = 0; sValid = 0; }
    __syncthreads();
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int c0 = 0; c0 < npad; c0 += kFvThreads) {
        const int i = c0 + tid;
        const unsigned long long key = i < npad ? fvKey[i] : ~0ull;
        const bool valid = key != ~0ull;
        const bool first = valid && (i == 0 || (uint32_t)(fvKey[i - 1] >> 32) != (uint32_t)(key >> 32));
        if (valid) fvIdx[i] = (uint32_t)key;
        const unsigned long long b = __ballot(first);
        if (lane == 0) sCnt[wave] = __popcll(b);
        if (valid) atomicMax(&sValid, i + 1);
        __syncthreads();
        int off = sBase;
        for (int k = 0; k < wave; k++) off += sCnt[k];
        if (first) { const int a = off + __popcll(b & ((1ull << lane) - 1)); fvNodes[a] = (uint32_t)(key >> 32); fvOff[a] = i; }
        __syncthreads();
        if (tid == 0) { int t = sBase; for (int k = 0; k < kFvThreads / 64; k++) t += sCnt[k]; sBase = t; }
        __syncthreads();
    }
    if (tid == 0) { fvOff[sBase] = sValid; *nnOut = sBase; }
}

// After the first PoseOptimization (Tracking.cc:2489-2508) and the first loop of SearchLocalPoints (:2998-3010): every point the motion search
// matched has been seen in this frame (inliers by SearchLocalPoints, outliers by the discard loop); outliers and bad points leave the frame.
// Thread 0 also derives Frame::UpdatePoseMatrices (Frame.cc:522-528) of the optimised pose in Sophus' / Eigen's float arithmetic: Rcw =
// q.toRotationMatrix(), tcw, Ow = conj(q) * (-tcw) (quaternion _transformVector), as [Rcw9 | tcw3 | Ow3 | K4] for k_is_in_frustum.
__device__ __forceinline__ void after_motion_body(int c, const int32_t *idx, const uint8_t *outlierC, int32_t *featMp, const int32_t *mpObs, const uint8_t *mpBad,
                                                  uint8_t *seen, const float *K4, TrackBlock *blk, const int32_t *searchHeader) {
    if (c == 0) pose_matrices19(blk->Tout, K4, blk->pose19);
    if (c == 0 && searchHeader) { blk->spec[0] = searchHeader[0]; blk->spec[1] = searchHeader[1]; }   // (the next search clears the header)
    if (c >= blk->start[1]) return;
    const int i = idx[c], mp = featMp[i];
    seen[mp] = outlierC[c] ? 2 : 1;                        // 2: discarded as an outlier (k_track_frustum: its mbTrackInView may still be set from an earlier frame)
    if (outlierC[c]) { featMp[i] = -1; return; }
    if (mpObs[mp] > 0) atomicAdd(&blk->counters[0], 1);
    if (mpBad[mp]) featMp[i] = -1;
}
__global__ void k_track_after_motion(const int32_t *idx, const uint8_t *outlierC, int32_t *featMp, const int32_t *mpObs, const uint8_t *mpBad, uint8_t *seen,
                                     const float *K4, TrackBlock *blk, const int32_t *searchHeader) {
    after_motion_body(blockIdx.x * blockDim.x + threadIdx.x, idx, outlierC, featMp, mpObs, mpBad, seen, K4, blk, searchHeader);
}

// Frame::isInFrustum of the local points SearchLocalPoints' second loop evaluates (not seen in this frame, not bad), writing the skip flag the query
// builder reads as isBad; also clears the search's result header and keeps a copy of the frame's map-point vector
struct FrustumArgs {
    int nmp;
    int n;
    const int32_t *featMp;
    int32_t *mpMotion;
    const uint8_t *local;
    const uint8_t *seen;
    const uint8_t *bad;
    uint8_t *skip;
    int32_t *searchHeader;
    const float *pose;
    float minX;
    float minY;
    float maxX;
    float maxY;
    float logScaleFactor;
    int nLevels;
    float viewingCosLimit;
    const float *mpPos;
    const float *mpNormal;
    const float *mpMinDist;
    const float *mpMaxDist;
    uint8_t *inView;
    float *projX;
    float *projY;
    int32_t *scaleLevel;
    float *viewCosOut;
    float *trackDepth;
    const int32_t *mpObs;
    const float *scaleFactors;
    float th;
    int farPoints;
    float thFar;
    Query *q;
    const uint8_t *staleIn;      // RumiTrackPoints.stale_in_view / stale_proj (nullptr: none)
    const float *staleProj;
};
__device__ __forceinline__ void frustum_body(int i, const FrustumArgs &F) {
    const auto nmp = F.nmp;
    const auto n = F.n;
    const auto featMp = F.featMp;
    const auto mpMotion = F.mpMotion;
    const auto local = F.local;
    const auto seen = F.seen;
    const auto bad = F.bad;
    const auto skip = F.skip;
    const auto searchHeader = F.searchHeader;
    const auto pose = F.pose;
    const auto minX = F.minX;
    const auto minY = F.minY;
    const auto maxX = F.maxX;
    const auto maxY = F.maxY;
    const auto logScaleFactor = F.logScaleFactor;
    const auto nLevels = F.nLevels;
    const auto viewingCosLimit = F.viewingCosLimit;
    const auto mpPos = F.mpPos;
    const auto mpNormal = F.mpNormal;
    const auto mpMinDist = F.mpMinDist;
    const auto mpMaxDist = F.mpMaxDist;
    const auto inView = F.inView;
    const auto projX = F.projX;
    const auto projY = F.projY;
    const auto scaleLevel = F.scaleLevel;
    const auto viewCosOut = F.viewCosOut;
    const auto trackDepth = F.trackDepth;
    const auto mpObs = F.mpObs;
    const auto scaleFactors = F.scaleFactors;
    const auto th = F.th;
    const auto farPoints = F.farPoints;
    const auto thFar = F.thFar;
    const auto q = F.q;
    if (i < 4) searchHeader[i] = 0;
    if (i < n) mpMotion[i] = featMp[i];                    // mvpMapPoints as TrackWithMotionModel leaves them (the local search may replace unobserved points)
    if (i >= nmp) return;
    // A discarded outlier (seen == 2) is not re-projected (mnLastFrameSeen == mnId) -- but a monocular frame's discard loop left its mbTrackInView
    // as an earlier frame set it (Nleft = -1, Tracking.cc:2489-2508), and SearchByProjection searches it at that OLD projection (ORBmatcher.cc:46-60)
    if (local[i] && !bad[i] && seen[i] == 2 && F.staleIn && F.staleIn[i]) {
        const float *sp = F.staleProj + (size_t)i * 5;
        skip[i] = 0;
        inView[i] = 2; projX[i] = sp[0]; projY[i] = sp[1]; scaleLevel[i] = (int)sp[2]; viewCosOut[i] = sp[3]; trackDepth[i] = sp[4];
        q[i] = mappoint_query(i, true, sp[0], sp[1], (int)sp[2], sp[3], sp[4], false, mpObs[i], scaleFactors, th, farPoints, thFar);
        return;
    }
    const uint8_t sk = !local[i] || seen[i] || bad[i];
    skip[i] = sk;
    // (the search's query of this point is built here too: k_queries_mappoints' work on the values at hand, one launch less)
    if (sk) {
        inView[i] = 0; projX[i] = -1; projY[i] = -1; scaleLevel[i] = 0; viewCosOut[i] = 0; trackDepth[i] = 0;
        q[i] = mappoint_query(i, false, -1.f, -1.f, 0, 0.f, 0.f, true, mpObs[i], scaleFactors, th, farPoints, thFar);
        return;
    }
    const float *R = pose, *t = pose + 9, *Ow = pose + 12, *K = pose + 15;
    const float *P = mpPos + (size_t)i * 3;
    uint8_t in = 0;
    float px = -1, py = -1, vc = 0, depth = 0;
    int lvl = 0;
    float Pc[3];
#pragma unroll
    for (int r = 0; r < 3; r++) Pc[r] = ((R[r * 3] * P[0] + R[r * 3 + 1] * P[1]) + R[r * 3 + 2] * P[2]) + t[r];
    const float Pc_dist = sqrtf((Pc[0] * Pc[0] + Pc[1] * Pc[1]) + Pc[2] * Pc[2]);
    if (!(Pc[2] < 0.0f)) {
        const float u = K[0] * Pc[0] / Pc[2] + K[2], v = K[1] * Pc[1] / Pc[2] + K[3];
        if (!(u < minX || u > maxX) && !(v < minY || v > maxY)) {
            px = u; py = v;
            const float maxD = 1.2f * mpMaxDist[i], minD = 0.8f * mpMinDist[i];
            const float P0 = P[0] - Ow[0], P1 = P[1] - Ow[1], P2 = P[2] - Ow[2];
            const float dist = sqrtf((P0 * P0 + P1 * P1) + P2 * P2);
            if (!(dist < minD || dist > maxD)) {
                const float *Pn = mpNormal + (size_t)i * 3;
                const float viewCos = ((P0 * Pn[0] + P1 * Pn[1]) + P2 * Pn[2]) / dist;
                if (!(viewCos < viewingCosLimit)) {
                    lvl = predict_scale(mpMaxDist[i], dist, logScaleFactor, nLevels);
                    in = 1; depth = Pc_dist; vc = viewCos;
                }
            }
        }
    }
    inView[i] = in; projX[i] = px; projY[i] = py; scaleLevel[i] = lvl; viewCosOut[i] = vc; trackDepth[i] = depth;
    q[i] = mappoint_query(i, in != 0, px, py, lvl, vc, depth, false, mpObs[i], scaleFactors, th, farPoints, thFar);
}

__global__ void k_track_frustum(FrustumArgs F) { frustum_body(blockIdx.x * blockDim.x + threadIdx.x, F); }

// after the last PoseOptimization: mvpMapPoints and mvbOutlier per feature into the result block, mnMatchesInliers (Tracking.cc:2573-2586)
__global__ void k_track_finish(const int32_t *idx, const uint8_t *outlierC, const int32_t *featMp, const int32_t *mpObs, uint8_t *outlierF, int32_t *mpOut,
                               int countInliers, TrackBlock *blk, const int32_t *searchHeader) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && searchHeader) { blk->spec[2] = searchHeader[0]; blk->spec[3] = searchHeader[1]; }
    if (c >= blk->start[1]) return;
    const int i = idx[c];
    mpOut[i] = featMp[i];
    if (countInliers) {
        outlierF[i] = outlierC[c];
        if (!outlierC[c] && mpObs[featMp[i]] > 0) atomicAdd(&blk->counters[1], 1);
    }
}

}  // namespace rumi

#include "rumi_internal.h"
#include "rumi_orb.h"
#include "rumi_track.h"
#include "rumi_voc.h"

// Frame::UndistortKeyPoints / ComputeImageBounds (R/lib_src/Frame.cc:770-826): cv::undistortPoints(mat, mat, K, mDistCoef, cv::Mat(), mK) per point,
// in double, operation by operation as OpenCV 3.4's cvUndistortPointsInternal does it (not in the tree: restated from the published algorithm,
// parity unpinned; oracle/frame_oracle.cc is the CPU statement the tests compare with): normalise, 5 fixed-point iterations of the inverse
// radial-tangential model, project with P = K.  Terms that are zero for (k1, k2, p1, p2, k3) keep their place: 0 * r2 is not dropped.
struct UndistortArgs { double fx, fy, cx, cy, ifx, ify, k1, k2, p1, p2, k3; };
__host__ __device__ inline void undistort_point(const UndistortArgs &A, float u, float v, float *uo, float *vo) {
    double x = u, y = v;
    x = (x - A.cx) * A.ifx; y = (y - A.cy) * A.ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((A.k3 * r2 + A.k2) * r2 + A.k1) * r2);
        const double deltaX = 2 * A.p1 * x * y + A.p2 * (r2 + 2 * x * x) + 0 * r2 + 0 * r2 * r2;
        const double deltaY = A.p1 * (r2 + 2 * y * y) + 2 * A.p2 * x * y + 0 * r2 + 0 * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = A.fx * x + 0 * y + A.cx, yy = 0 * x + A.fy * y + A.cy, ww = 1. / (0 * x + 0 * y + 1);
    *uo = (float)(xx * ww); *vo = (float)(yy * ww);
}
// mvKeysUn: the extractor's key-points with pt replaced (Frame.cc:791-796); the count is read where the extractor left it (nDev) or given (n)
__global__ void k_undistort_keys(const int32_t *nDev, int n, const RumiKeyPoint *__restrict__ keys, RumiKeyPoint *__restrict__ keysUn, UndistortArgs A) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (nDev ? *nDev : n)) return;
    RumiKeyPoint k = keys[i];
    undistort_point(A, k.x, k.y, &k.x, &k.y);
    keysUn[i] = k;
}

struct RumiTracker {
    int device = 0, cap = 0, maxPts = 0, nlevels = 0;
    RumiOrbConfig cfg{};
    RumiOrb *ext = nullptr;
    RumiMatcher *m = nullptr;
    uint8_t *dImage = nullptr; size_t imageBytes = 0;
    hipStream_t upStream = nullptr;      // the step's uploads travel beside the extraction (rumi_track_frame)
    hipEvent_t evUp = nullptr;
    uint8_t *hImage = nullptr;           // pinned staging of the caller's (pageable) image: a plain memcpy + one asynchronous copy (the runtime's own
                                         // staging of a pageable source serialises the call for ~0.1 ms)
    // ONE device block [TrackBlock | mp cap*4 | mp after the motion model cap*4 | outlier cap | in_view maxPts | record 8 + 60 cap] and its pinned mirror: one copy brings a frame's results back
    uint8_t *dBlk = nullptr, *hBlk = nullptr; size_t oMp = 0, oMpM = 0, oOut = 0, oView = 0, oRec = 0, blkBytes = 0, recordBytes = 0;
    float *dInvSigma2 = nullptr, *dXw = nullptr, *dObs = nullptr, *dW = nullptr;
    int32_t *dIdx = nullptr;
    uint8_t *dOutC = nullptr, *dActive = nullptr, *dSeen = nullptr, *dBad = nullptr, *dLocal = nullptr, *dStaleIn = nullptr;
    float *dStaleProj = nullptr;
    size_t projN16 = 0; int projN = 0;       // the projection arrays the last SearchLocalPoints left in the matcher's staging block (rumi_track_last_projections)
    double *dChi = nullptr;
    float scale[64] = {0};
    // the step-wise entries (rumi_track_extract / _motion / _reference_keyframe / _local): the frame that is resident, and its BoW transform
    int curN = -1, curW = 0, curH = 0, curMono = -1;
    // lens distortion (rumi_track_set_distortion): mvKeysUn of the resident frame and the undistorted image bounds (mnMinX .. mnMaxY)
    bool distort = false;
    UndistortArgs ua{};
    RumiKeyPoint *dKeysUn = nullptr;
    float bounds[4] = {0, 0, 0, 0};
    // the frame's BoW transform: one block [weight f64 x cap | word u32 x cap | node u32 x cap] and its pinned mirror (one copy back)
    uint8_t *dBow = nullptr, *hBow = nullptr;
    uint32_t *dWord = nullptr, *dNode = nullptr; double *dWeight = nullptr; int32_t *dNN = nullptr;
};

namespace {
void track_bounds(RumiTracker *t, int w, int h, RumiFrameFeatures *F);
void track_undistort(RumiTracker *t);
}  // namespace

extern "C" void rumi_track_destroy(RumiTracker *t) {
    if (!t) return;
    (void)hipSetDevice(t->device);
    rumi_orb_destroy(t->ext);
    rumi_match_destroy(t->m);
    void *p[] = {t->dImage, t->dBlk, t->dInvSigma2, t->dXw, t->dObs, t->dW, t->dIdx, t->dOutC, t->dActive, t->dSeen, t->dBad, t->dLocal, t->dChi, t->dBow, t->dNN, t->dStaleIn, t->dStaleProj, t->dKeysUn};
    for (void *q : p) if (q) (void)hipFree(q);
    if (t->hBlk) (void)hipHostFree(t->hBlk);
    if (t->hBow) (void)hipHostFree(t->hBow);
    if (t->hImage) (void)hipHostFree(t->hImage);
    if (t->upStream) (void)hipStreamDestroy(t->upStream);
    if (t->evUp) (void)hipEventDestroy(t->evUp);
    delete t;
}

extern "C" int rumi_track_create(const RumiOrbConfig *cfg, int32_t max_points, int32_t device, RumiTracker **out) {
    if (!out) return RUMI_E_INVALID;
    *out = nullptr;
    if (!cfg || max_points < 1 || cfg->nlevels < 1 || cfg->nlevels > 16) return RUMI_E_INVALID;
    RumiTracker *t = new RumiTracker();
    t->cfg = *cfg; t->cfg.max_batch = 1; t->cfg.device = device;
    t->nlevels = cfg->nlevels;
    t->cap = cfg->nfeatures + 4 * cfg->nlevels + 64;                // what the facade's ORBextractor::operator() reserves
    t->maxPts = max_points;
    int rc = rumi_orb_create(&t->cfg, &t->ext);
    if (rc == RUMI_OK) rc = rumi_match_create(t->cap, std::max(max_points, t->cap), device, &t->m);
    if (rc != RUMI_OK) { rumi_track_destroy(t); return rc; }
    t->device = t->m->device;
    const size_t C = t->cap, P = max_points;
    auto al = [](size_t x) { return (x + 63) & ~(size_t)63; };
    t->imageBytes = (size_t)((cfg->max_width + 3) & ~3) * cfg->max_height;
    t->recordBytes = 8 + 60 * C;
    t->oMp = al(sizeof(TrackBlock)); t->oMpM = al(t->oMp + C * 4); t->oOut = al(t->oMpM + C * 4); t->oView = al(t->oOut + C); t->oRec = al(t->oView + P); t->blkBytes = al(t->oRec + t->recordBytes);
#define TRYA(x) if ((rc = (x)) != RUMI_OK) { rumi_track_destroy(t); return rc; }
    TRYA(dalloc(&t->dImage, t->imageBytes + 64)); TRYA(dalloc(&t->dBlk, t->blkBytes)); TRYA(dalloc(&t->dInvSigma2, 64));
    TRYA(dalloc(&t->dXw, C * 3)); TRYA(dalloc(&t->dObs, C * 2)); TRYA(dalloc(&t->dW, C)); TRYA(dalloc(&t->dIdx, C));
    TRYA(dalloc(&t->dOutC, C)); TRYA(dalloc(&t->dActive, C)); TRYA(dalloc(&t->dSeen, P)); TRYA(dalloc(&t->dBad, P)); TRYA(dalloc(&t->dLocal, P)); TRYA(dalloc(&t->dStaleIn, P)); TRYA(dalloc(&t->dStaleProj, P * 5)); TRYA(dalloc(&t->dChi, C));
    TRYA(dalloc(&t->dBow, C * 16)); TRYA(dalloc(&t->dNN, 4));
    t->dWeight = reinterpret_cast<double *>(t->dBow); t->dWord = reinterpret_cast<uint32_t *>(t->dBow + C * 8); t->dNode = t->dWord + C;
#undef TRYA
    if (hipHostMalloc((void **)&t->hBlk, t->blkBytes, hipHostMallocDefault) != hipSuccess || hipHostMalloc((void **)&t->hBow, C * 16, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&t->hImage, t->imageBytes + 64, hipHostMallocDefault) != hipSuccess ||
        hipStreamCreateWithFlags(&t->upStream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&t->evUp, hipEventDisableTiming) != hipSuccess) {
        rumi_track_destroy(t); return RUMI_E_NO_DEVICE;
    }
    float inv2[64] = {0};
    rumi_orb_tables(cfg, t->scale, nullptr, nullptr, inv2, nullptr, nullptr);
    if (hipMemcpy(t->dInvSigma2, inv2, sizeof(inv2), hipMemcpyHostToDevice) != hipSuccess) { rumi_track_destroy(t); return RUMI_E_NO_DEVICE; }
    *out = t;
    return RUMI_OK;
}

extern "C" int rumi_track_frame(RumiTracker *t, const uint8_t *img, int32_t w, int32_t h, int32_t stride, const float *K4, const float *Tcw_pred7,
                                const RumiKeyPoint *last_keys_un, int32_t nlast, const int32_t *last_mp, const uint8_t *last_outlier,
                                const RumiTrackPoints *pts, float th_motion, float th_local, int32_t far_points, float th_far_points,
                                RumiKeyPoint *keys_out, uint8_t *desc_out, int32_t cap, int32_t *frame_mp_motion, int32_t *frame_mp, uint8_t *outlier,
                                uint8_t *in_view, RumiTrackResult *res) {
    if (!t || !img || !K4 || !Tcw_pred7 || !pts || !res || !keys_out || !desc_out || !frame_mp_motion || !frame_mp || !outlier || nlast < 0 || pts->n < 0 ||
        stride < w || (nlast > 0 && (!last_keys_un || !last_mp || !last_outlier)) ||
        (pts->n > 0 && (!pts->pos || !pts->normal || !pts->min_dist || !pts->max_dist || !pts->desc || !pts->obs || !pts->bad || !pts->local || !in_view)))
        return RUMI_E_INVALID;
    if (w <= 0 || h <= 0) return RUMI_E_EMPTY;
    if (w > t->cfg.max_width || h > t->cfg.max_height) { g_lastError = "rumi_track_frame: image larger than the tracker was created for"; return RUMI_E_CAPACITY; }
    t->curN = -1;
    RumiMatcher *m = t->m;
    const int nmp = pts->n;
    if (nlast > m->maxQ || nmp > t->maxPts || cap < t->cap) { g_lastError = "rumi_track_frame: more points / features than the tracker was created for, or cap too small"; return RUMI_E_CAPACITY; }
    for (int i = 0; i < nlast; i++) if (last_mp[i] >= nmp) { g_lastError = "rumi_track_frame: last_mp index outside the point table"; return RUMI_E_INVALID; }
    HIP_TRY(hipSetDevice(t->device));
    std::memset(res, 0, sizeof(*res));
    t->projN = 0;
    res->mono_index = -1; res->th_motion = (int32_t)th_motion;
    TrackBlock *dB = reinterpret_cast<TrackBlock *>(t->dBlk);
    int32_t *dMpOut = reinterpret_cast<int32_t *>(t->dBlk + t->oMp), *dMpMotion = reinterpret_cast<int32_t *>(t->dBlk + t->oMpM);
    uint8_t *dOutF = t->dBlk + t->oOut, *dView = t->dBlk + t->oView, *dRecord = t->dBlk + t->oRec;

    // ---- stage 1: ORBextractor::operator() on the device; only the two counts come back (launch sizes need n)
    const int wp = (w + 3) & ~3;
    if (!(img == t->hImage && stride == wp))                // (a caller that captured straight into rumi_track_image_buffer's memory has nothing to stage)
        for (int y = 0; y < h; y++) std::memcpy(t->hImage + (size_t)y * wp, img + (size_t)y * stride, (size_t)w);      // image -> pinned -> device (async)
    HIP_TRY(hipMemcpyAsync(t->dImage, t->hImage, (size_t)wp * h, hipMemcpyHostToDevice, nullptr));
    int rc = rumi_orb_extract_batch_records_async(t->ext, t->dImage, 1, w, h, wp, (int64_t)wp * h, 0, 1000, dRecord, (int64_t)t->recordBytes, t->cap, nullptr);
    if (rc != RUMI_OK) return rc;
    track_undistort(t);
    // ---- uploads of the whole step: one pinned block, one copy, scattered on the device; the frame itself is read where the extractor left it.
    // None of it depends on the extraction: the host fills the block while the extraction runs, and only then waits for the two counts.
    RumiFrameFeatures F{};
    F.n = 0;
    F.nlevels = t->nlevels; F.scale_factors = t->scale;
    track_bounds(t, w, h, &F);
    FrameDev fd;
    if ((rc = upload_frame(m, &F, &fd)) != RUMI_OK) return rc;
    float pose[11];
    std::memcpy(pose, Tcw_pred7, 7 * sizeof(float)); std::memcpy(pose + 7, K4, 4 * sizeof(float));
    H2D(m->dPose, pose, 11);
    if (nmp > 0) {
        H2D(m->dF[0], pts->pos, (size_t)nmp * 3); H2D(m->dF[1], pts->normal, (size_t)nmp * 3); H2D(m->dF[2], pts->min_dist, nmp); H2D(m->dF[3], pts->max_dist, nmp);
        H2D(m->dI[1], pts->obs, nmp); H2D(m->dQDesc, pts->desc, (size_t)nmp * 32); H2D(t->dBad, pts->bad, nmp); H2D(t->dLocal, pts->local, nmp);
        if (pts->stale_in_view && pts->stale_proj) { H2D(t->dStaleIn, pts->stale_in_view, nmp); H2D(t->dStaleProj, pts->stale_proj, (size_t)nmp * 5); }
    }
    if (nlast > 0) { H2D(m->dQKeys, last_keys_un, nlast); H2D(m->dI[0], last_mp, nlast); H2D(m->dU8a, last_outlier, nlast); }
    const bool gridWanted = m->gridPending;                 // (the grid needs the feature count: it is built below)
    m->gridPending = false;
    static const int envSpec = std::getenv("RUMI_TRACK_SPECULATE") ? std::atoi(std::getenv("RUMI_TRACK_SPECULATE")) : 1;
    static const bool noFusedLists = std::getenv("RUMI_MATCH_NO_FUSED") != nullptr;
    const size_t n16 = ((size_t)nmp + 15) & ~(size_t)15;
    const float logSf = std::log(t->cfg.scale_factor);
    const bool canSpec = envSpec && !noFusedLists && nlast > 0 && nmp > 0 && m->listCap / (size_t)std::max(nlast, nmp) >= 64 && n16 * 21 <= m->stageCap;
    const int gC = std::max(1, (t->cap + 255) / 256);
    m->upStream = t->upStream;                              // the copy and the scatter, on a stream of their own beside the extraction
    const int rcUp = flush_uploads(m);
    m->upStream = nullptr;
    if (rcUp != RUMI_OK) return rcUp;
    if (canSpec) {
        // what the usual case (below) needs and the extraction does not feed: the cleared frame (sized by the capacity) and the motion-model queries
        hipLaunchKernelGGL(k_track_init, dim3(std::max(1, (std::max(t->cap, nmp) + 255) / 256)), dim3(256), 0, t->upStream, t->cap, nmp, 1, m->dFeatMp, m->dOut, t->dSeen, dOutF,
                           dMpOut, m->dPose, dB);
        hipLaunchKernelGGL(k_queries_frame, dim3((nlast + 255) / 256), dim3(256), 0, t->upStream, nlast, m->dQKeys, m->dI[0], m->dU8a, m->dF[0], m->dI[1], m->dPose,
                           m->dPose + 7, m->dScale, th_motion, fd.minX, fd.minY, fd.maxX, fd.maxY, m->dQ);
    }
    HIP_TRY(hipEventRecord(t->evUp, t->upStream));
    HIP_TRY(hipStreamWaitEvent(nullptr, t->evUp, 0));      // (behind the extraction in the main queue: by then the event has long fired)
    // ---- the usual case in ONE queue, no host round trip: the first search finds >= 20 matches and no candidate list overflows.  Every launch
    // of stages 2-5 goes out back to back behind the extraction -- the feature count and the searches' counts stay on the device (launches are
    // sized by their upper bounds), the searches' result headers are kept in the block -- the block comes back once, and only if a header says
    // otherwise (fewer than 20 matches: the 2 th retry; a list overflow; no key-point at all) the step is redone stage by stage.
    int n = t->cap;                                          // an upper bound until the two counts have been read
    auto take_counts = [&](const int32_t *counts) {
        n = counts[0];
        res->n = n; res->mono_index = counts[1];
        t->curN = n; t->curW = w; t->curH = h; t->curMono = counts[1];      // the frame is resident for the step-wise entries too
    };
    if (!canSpec) {
        int32_t counts[2] = {0, -1};
        HIP_TRY(hipMemcpy(counts, dRecord, 8, hipMemcpyDeviceToHost));
        if ((rc = rumi_orb_sync(t->ext)) != RUMI_OK) return rc;
        take_counts(counts);
    }
    const RumiKeyPoint *dKp = t->distort ? t->dKeysUn : reinterpret_cast<const RumiKeyPoint *>(dRecord + 8);      // mvKeysUn: what every stage below reads
    const uint8_t *dDs = dRecord + 8 + (size_t)t->cap * sizeof(RumiKeyPoint);
    fd.n = n; fd.keys = dKp; fd.desc = dDs;
    m->gridN = n; m->gridKeys = dKp; m->gridNDev = canSpec ? reinterpret_cast<const int32_t *>(dRecord) : nullptr;
    m->gridPending = gridWanted;
    FLUSH(m);                                               // the grid of the resident frame
    int gI = std::max(1, (std::max(std::max(n, nmp), 4) + 255) / 256);
    if (!canSpec) hipLaunchKernelGGL(k_track_init, dim3(gI), dim3(256), 0, nullptr, n, nmp, 1, m->dFeatMp, m->dOut, t->dSeen, dOutF, dMpOut, m->dPose, dB);
    bool small = n <= kTrackLdsEdges;                       // the frame's correspondences fit the LDS instantiation of k_pose_opt for sure
    bool done = false;
    if (canSpec) {
        auto search = [&](int mode, int nq, float nnratio, int checkOri) -> int {
            const int rcl = build_lists(m, mode, nq, fd, m->dQDesc, false, true);
            if (rcl != RUMI_OK) return rcl;
            // (the search ends with the gather of PoseOptimization's correspondences: k_resolve's tail)
            ResolveArgs A{mode, nq, fd.n, m->dQ, m->dCounts, m->dOffsets, m->dLists, fd.keys, m->dI[1], m->dFeatMp, m->dAssign, m->dNmatches,
                          nnratio, checkOri, nullptr, 0.f, 0, m->dOverflow, m->dF[0], t->dInvSigma2, t->dXw, t->dObs, t->dW, t->dIdx, dB->start, nullptr};
            hipLaunchKernelGGL(k_resolve, dim3(1), dim3(1024), (size_t)std::max(fd.n, 1) * sizeof(int32_t), nullptr, A);
            return RUMI_OK;
        };
        if ((rc = search(MODE_FRAME, nlast, 0.f, 1)) != RUMI_OK) return rc;      // (its queries were built beside the extraction, above)
        if ((rc = rumi::pose_opt_device(dB->start, t->dXw, t->dObs, t->dW, m->dPose + 7, m->dPose, dB->Tout, t->dOutC, dB->nGood, t->dActive, t->dChi, small, nullptr)) != RUMI_OK) return rc;
        uint8_t *dSkip = m->dU8b;
        float *dX = reinterpret_cast<float *>(m->dStage + n16), *dY = dX + n16, *dC = dY + n16, *dD = dC + n16;
        int32_t *dL = reinterpret_cast<int32_t *>(dD + n16);
        { const FrustumArgs FA{nmp, n, m->dFeatMp, dMpMotion, t->dLocal, t->dSeen, t->dBad, dSkip, m->dOut, dB->pose19, fd.minX,
                           fd.minY, fd.maxX, fd.maxY, logSf, t->nlevels, 0.5f, m->dF[0], m->dF[1], m->dF[2], m->dF[3], dView, dX, dY, dL, dC, dD,
                           m->dI[1], m->dScale, th_local, far_points, th_far_points, m->dQ,
                           (pts->stale_in_view && pts->stale_proj) ? t->dStaleIn : nullptr, (pts->stale_in_view && pts->stale_proj) ? t->dStaleProj : nullptr};
          hipLaunchKernelGGL(k_track_after_motion, dim3(gC), dim3(256), 0, nullptr, t->dIdx, t->dOutC, m->dFeatMp, m->dI[1], t->dBad, t->dSeen, m->dPose + 7, dB, (const int32_t *)m->dOut);
          // (both as ONE 1024-thread workgroup -- the frustum test reads the seen flags and the pose matrices the first half writes -- measured: 0.427-0.436 ms
          // against 0.430-0.431 for the frame, no gain; not kept)
          hipLaunchKernelGGL(k_track_frustum, dim3(gI), dim3(256), 0, nullptr, FA); t->projN16 = n16; t->projN = nmp; }
        if ((rc = search(MODE_MAPPOINTS, nmp, 0.8f, 0)) != RUMI_OK) return rc;
        if ((rc = rumi::pose_opt_device(dB->start, t->dXw, t->dObs, t->dW, m->dPose + 7, dB->Tout, dB->Tout + 7, t->dOutC, dB->nGood + 1, t->dActive, t->dChi, small, nullptr)) != RUMI_OK) return rc;
        hipLaunchKernelGGL(k_track_finish, dim3(gC), dim3(256), 0, nullptr, t->dIdx, t->dOutC, m->dFeatMp, m->dI[1], dOutF, dMpOut, 1, dB, (const int32_t *)m->dOut);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(t->hBlk, t->dBlk, t->oRec + 8 + (size_t)t->cap * sizeof(RumiKeyPoint) + (size_t)n * 32, hipMemcpyDeviceToHost));
        if ((rc = rumi_orb_sync(t->ext)) != RUMI_OK) return rc;
        take_counts(reinterpret_cast<const int32_t *>(t->hBlk + t->oRec));
        const TrackBlock *hS = reinterpret_cast<const TrackBlock *>(t->hBlk);
        if (n > 0 && hS->spec[0] >= 20 && hS->spec[1] == 0 && hS->spec[3] == 0) {
            res->nmatches_motion = hS->spec[0];
            res->nmatches_local = hS->spec[2];
            done = true;
        } else {
            // not the usual case: start over from the cleared frame (the staged inputs and the frame's grid are still on the device)
            fd.n = n; m->gridN = n;
            gI = std::max(1, (std::max(std::max(n, nmp), 4) + 255) / 256);
            small = n <= kTrackLdsEdges;
            hipLaunchKernelGGL(k_track_init, dim3(gI), dim3(256), 0, nullptr, n, nmp, 1, m->dFeatMp, m->dOut, t->dSeen, dOutF, dMpOut, m->dPose, dB);
        }
    }
    bool localRan = done;
    if (!done) {
    // ---- stage 2: SearchByProjection(Cur, Last, th, mono), once more with 2 * th below 20 matches (Tracking.cc:2466-2474)
    int nm = 0;
    std::vector<int32_t> tmpMp((size_t)std::max(n, 1));
    for (int attempt = 0; attempt < 2 && n > 0 && nlast > 0 && nmp > 0; attempt++) {
        const float th = attempt == 0 ? th_motion : 2 * th_motion;
        if (attempt == 1) hipLaunchKernelGGL(k_track_init, dim3(gI), dim3(256), 0, nullptr, n, nmp, 0, m->dFeatMp, m->dOut, t->dSeen, dOutF, dMpOut, m->dPose, dB);
        hipLaunchKernelGGL(k_queries_frame, dim3((nlast + 255) / 256), dim3(256), 0, nullptr, nlast, m->dQKeys, m->dI[0], m->dU8a, m->dF[0], m->dI[1], m->dPose,
                           m->dPose + 7, m->dScale, th, fd.minX, fd.minY, fd.maxX, fd.maxY, m->dQ);
        if ((rc = run_search(m, MODE_FRAME, nlast, fd, m->dQDesc, m->dI[1], 0.f, 1, tmpMp.data(), &nm)) != RUMI_OK) return rc;
        res->th_motion = (int32_t)th;
        if (nm >= 20) break;
    }
    res->nmatches_motion = nm;
    if (nm >= 20) {
        // ---- stage 3: PoseOptimization on the matches, outliers leave the frame
        hipLaunchKernelGGL(k_track_gather, dim3(1), dim3(1024), 0, nullptr, n, dKp, m->dFeatMp, m->dF[0], t->dInvSigma2, t->dXw, t->dObs, t->dW, t->dIdx, dB->start);
        if ((rc = rumi::pose_opt_device(dB->start, t->dXw, t->dObs, t->dW, m->dPose + 7, m->dPose, dB->Tout, t->dOutC, dB->nGood, t->dActive, t->dChi, small, nullptr)) != RUMI_OK) return rc;
        hipLaunchKernelGGL(k_track_after_motion, dim3(gC), dim3(256), 0, nullptr, t->dIdx, t->dOutC, m->dFeatMp, m->dI[1], t->dBad, t->dSeen, m->dPose + 7, dB, (const int32_t *)nullptr);
        // ---- stage 4: SearchLocalPoints with the optimised pose
        if (n16 * 21 > m->stageCap) { g_lastError = "rumi_track_frame: point table exceeds the staging block"; return RUMI_E_CAPACITY; }
        uint8_t *dSkip = m->dU8b;
        float *dX = reinterpret_cast<float *>(m->dStage + n16), *dY = dX + n16, *dC = dY + n16, *dD = dC + n16;
        int32_t *dL = reinterpret_cast<int32_t *>(dD + n16);
        { const FrustumArgs FA{nmp, n, m->dFeatMp, dMpMotion, t->dLocal, t->dSeen, t->dBad, dSkip, m->dOut, dB->pose19, fd.minX,
                           fd.minY, fd.maxX, fd.maxY, logSf, t->nlevels, 0.5f, m->dF[0], m->dF[1], m->dF[2], m->dF[3], dView, dX, dY, dL, dC, dD,
                           m->dI[1], m->dScale, th_local, far_points, th_far_points, m->dQ,
                           (pts->stale_in_view && pts->stale_proj) ? t->dStaleIn : nullptr, (pts->stale_in_view && pts->stale_proj) ? t->dStaleProj : nullptr};
          hipLaunchKernelGGL(k_track_frustum, dim3(gI), dim3(256), 0, nullptr, FA); t->projN16 = n16; t->projN = nmp; }
        int nmLocal = 0;
        if ((rc = run_search(m, MODE_MAPPOINTS, nmp, fd, m->dQDesc, m->dI[1], 0.8f, 0, tmpMp.data(), &nmLocal)) != RUMI_OK) return rc;
        res->nmatches_local = nmLocal;
        localRan = true;
        // ---- stage 5: PoseOptimization on everything the frame now holds
        hipLaunchKernelGGL(k_track_gather, dim3(1), dim3(1024), 0, nullptr, n, dKp, m->dFeatMp, m->dF[0], t->dInvSigma2, t->dXw, t->dObs, t->dW, t->dIdx, dB->start);
        if ((rc = rumi::pose_opt_device(dB->start, t->dXw, t->dObs, t->dW, m->dPose + 7, dB->Tout, dB->Tout + 7, t->dOutC, dB->nGood + 1, t->dActive, t->dChi, small, nullptr)) != RUMI_OK) return rc;
        hipLaunchKernelGGL(k_track_finish, dim3(gC), dim3(256), 0, nullptr, t->dIdx, t->dOutC, m->dFeatMp, m->dI[1], dOutF, dMpOut, 1, dB, (const int32_t *)nullptr);
    } else if (nm > 0) {                                       // fewer than 20 matches: the frame keeps them (the caller falls back to TrackReferenceKeyFrame)
        hipLaunchKernelGGL(k_track_gather, dim3(1), dim3(1024), 0, nullptr, n, dKp, m->dFeatMp, m->dF[0], t->dInvSigma2, t->dXw, t->dObs, t->dW, t->dIdx, dB->start);
        hipLaunchKernelGGL(k_track_finish, dim3(gC), dim3(256), 0, nullptr, t->dIdx, t->dOutC, m->dFeatMp, m->dI[1], dOutF, dMpOut, 0, dB, (const int32_t *)nullptr);
    }
    // ---- one copy back: header, mvpMapPoints, mvbOutlier, mbTrackInView and the extractor's record
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(t->hBlk, t->dBlk, t->oRec + 8 + (size_t)t->cap * sizeof(RumiKeyPoint) + (size_t)n * 32, hipMemcpyDeviceToHost));
    }
    const TrackBlock *hB = reinterpret_cast<const TrackBlock *>(t->hBlk);
    std::memcpy(res->Tcw_motion, hB->Tout, 28); std::memcpy(res->Tcw, hB->Tout + 7, 28);
    std::memcpy(res->Rcw, hB->pose19, 36); std::memcpy(res->tcw, hB->pose19 + 9, 12); std::memcpy(res->Ow, hB->pose19 + 12, 12);
    res->ngood_motion = hB->nGood[0]; res->ngood_local = hB->nGood[1]; res->nmatches_map = hB->counters[0]; res->matches_inliers = hB->counters[1];
    if (n > 0) {
        std::memcpy(frame_mp, t->hBlk + t->oMp, (size_t)n * 4); std::memcpy(outlier, t->hBlk + t->oOut, (size_t)n);
        std::memcpy(keys_out, t->hBlk + t->oRec + 8, (size_t)n * sizeof(RumiKeyPoint));
        std::memcpy(desc_out, t->hBlk + t->oRec + 8 + (size_t)t->cap * sizeof(RumiKeyPoint), (size_t)n * 32);
    }
    int nTo = 0;
    if (nmp > 0) {
        if (localRan) std::memcpy(in_view, t->hBlk + t->oView, (size_t)nmp); else std::memset(in_view, 0, (size_t)nmp);
        for (int j = 0; j < nmp; j++) nTo += in_view[j] == 1;         // (2: a stale flag of an earlier frame, not an isInFrustum of this one)
    }
    res->n_to_match = nTo;
    if (n > 0) std::memcpy(frame_mp_motion, localRan ? t->hBlk + t->oMpM : t->hBlk + t->oMp, (size_t)n * 4);
    return RUMI_OK;
}

// ==================================================================================================================
// The same stages one member function of Tracking at a time (include/rumi_track.h, "step-wise entries"): the frame extracted by
// rumi_track_extract stays on the device -- key-points, descriptors, grid, FeatureVector -- while the host runs the reference's own control
// flow between the calls (the decisions of TrackWithMotionModel / TrackReferenceKeyFrame, UpdateLocalMap).
// ==================================================================================================================
namespace {
// Frame::ComputeImageBounds (Frame.cc:799-826): the image rectangle, or the undistorted corners' hull with lens distortion
void track_bounds(RumiTracker *t, int w, int h, RumiFrameFeatures *F) {
    if (!t->distort) { t->bounds[0] = 0; t->bounds[1] = 0; t->bounds[2] = (float)w; t->bounds[3] = (float)h; }
    else {
        const float c[8] = {0, 0, (float)w, 0, 0, (float)h, (float)w, (float)h};
        float u[8];
        for (int i = 0; i < 4; i++) undistort_point(t->ua, c[2 * i], c[2 * i + 1], &u[2 * i], &u[2 * i + 1]);
        t->bounds[0] = std::min(u[0], u[4]); t->bounds[2] = std::max(u[2], u[6]);
        t->bounds[1] = std::min(u[1], u[3]); t->bounds[3] = std::max(u[5], u[7]);
    }
    F->min_x = t->bounds[0]; F->min_y = t->bounds[1]; F->max_x = t->bounds[2]; F->max_y = t->bounds[3];
}
// mvKeysUn of the frame the extractor has just been asked for (same queue, behind the extraction)
void track_undistort(RumiTracker *t) {
    if (!t->distort) return;
    const uint8_t *dRecord = t->dBlk + t->oRec;
    hipLaunchKernelGGL(k_undistort_keys, dim3((t->cap + 255) / 256), dim3(256), 0, nullptr, reinterpret_cast<const int32_t *>(dRecord), t->cap,
                       reinterpret_cast<const RumiKeyPoint *>(dRecord + 8), t->dKeysUn, t->ua);
}
// the resident frame as the matcher's kernels address it (upload_frame of an empty frame queues the scale table and the cleared result header)
int track_frame_dev(RumiTracker *t, FrameDev *fd) {
    RumiMatcher *m = t->m;
    RumiFrameFeatures F{};
    F.n = 0; F.nlevels = t->nlevels; F.scale_factors = t->scale;
    track_bounds(t, t->curW, t->curH, &F);
    const int rc = upload_frame(m, &F, fd);
    if (rc != RUMI_OK) return rc;
    const uint8_t *dRecord = t->dBlk + t->oRec;
    fd->n = t->curN;
    fd->keys = t->distort ? t->dKeysUn : reinterpret_cast<const RumiKeyPoint *>(dRecord + 8);
    fd->desc = dRecord + 8 + (size_t)t->cap * sizeof(RumiKeyPoint);
    m->gridN = t->curN; m->gridKeys = fd->keys;
    return RUMI_OK;
}
int track_check_points(const RumiTrackPoints *pts, bool needFrustum) {
    if (!pts || pts->n < 0) return RUMI_E_INVALID;
    if (pts->n > 0 && (!pts->pos || !pts->desc || !pts->obs || !pts->bad)) return RUMI_E_INVALID;
    if (pts->n > 0 && needFrustum && (!pts->normal || !pts->min_dist || !pts->max_dist || !pts->local)) return RUMI_E_INVALID;
    return RUMI_OK;
}
}  // namespace

// The tracker's pinned staging buffer for a w x h frame, for a caller that lets its camera driver / decoder write the frame there (e.g. a
// cv::Mat constructed on this memory): rumi_track_frame / rumi_track_extract called with this pointer and stride skip their staging copy.
extern "C" int rumi_track_image_buffer(RumiTracker *t, int32_t w, int32_t h, uint8_t **buf, int32_t *stride) {
    if (!t || !buf || !stride) return RUMI_E_INVALID;
    if (w <= 0 || h <= 0 || w > t->cfg.max_width || h > t->cfg.max_height) { g_lastError = "rumi_track_image_buffer: frame larger than the tracker was created for"; return RUMI_E_CAPACITY; }
    *buf = t->hImage; *stride = (w + 3) & ~3;
    return RUMI_OK;
}

extern "C" int rumi_track_extract(RumiTracker *t, const uint8_t *img, int32_t w, int32_t h, int32_t stride, RumiKeyPoint *keys_out, uint8_t *desc_out,
                                  int32_t cap, int32_t *n_out, int32_t *mono_out) {
    if (!t || !img || !keys_out || !desc_out || !n_out || !mono_out || stride < w) return RUMI_E_INVALID;
    *n_out = 0; *mono_out = -1;
    if (w <= 0 || h <= 0) return RUMI_E_EMPTY;
    if (w > t->cfg.max_width || h > t->cfg.max_height || cap < t->cap) { g_lastError = "rumi_track_extract: image larger than the tracker was created for, or cap too small"; return RUMI_E_CAPACITY; }
    HIP_TRY(hipSetDevice(t->device));
    t->curN = -1;
    uint8_t *dRecord = t->dBlk + t->oRec;
    const int wp = (w + 3) & ~3;
    if (!(img == t->hImage && stride == wp))                // (a caller that captured straight into rumi_track_image_buffer's memory has nothing to stage)
        for (int y = 0; y < h; y++) std::memcpy(t->hImage + (size_t)y * wp, img + (size_t)y * stride, (size_t)w);      // image -> pinned -> device (async)
    HIP_TRY(hipMemcpyAsync(t->dImage, t->hImage, (size_t)wp * h, hipMemcpyHostToDevice, nullptr));
    int rc = rumi_orb_extract_batch_records_async(t->ext, t->dImage, 1, w, h, wp, (int64_t)wp * h, 0, 1000, dRecord, (int64_t)t->recordBytes, t->cap, nullptr);
    if (rc != RUMI_OK) return rc;
    track_undistort(t);
    int32_t counts[2] = {0, -1};
    HIP_TRY(hipMemcpy(counts, dRecord, 8, hipMemcpyDeviceToHost));
    if ((rc = rumi_orb_sync(t->ext)) != RUMI_OK) return rc;
    const int n = counts[0];
    if (n > 0) {
        HIP_TRY(hipMemcpyAsync(t->hBlk + t->oRec + 8, dRecord + 8, (size_t)n * sizeof(RumiKeyPoint), hipMemcpyDeviceToHost, nullptr));
        HIP_TRY(hipMemcpy(t->hBlk + t->oRec + 8 + (size_t)t->cap * sizeof(RumiKeyPoint), dRecord + 8 + (size_t)t->cap * sizeof(RumiKeyPoint), (size_t)n * 32, hipMemcpyDeviceToHost));
        std::memcpy(keys_out, t->hBlk + t->oRec + 8, (size_t)n * sizeof(RumiKeyPoint));
        std::memcpy(desc_out, t->hBlk + t->oRec + 8 + (size_t)t->cap * sizeof(RumiKeyPoint), (size_t)n * 32);
    }
    t->curN = n; t->curW = w; t->curH = h; t->curMono = counts[1];
    { RumiFrameFeatures Fb{}; track_bounds(t, w, h, &Fb); }    // mnMinX .. mnMaxY of this frame (rumi_track_undistorted)
    *n_out = n; *mono_out = counts[1];
    return RUMI_OK;
}

extern "C" int rumi_track_motion(RumiTracker *t, const float *K4, const float *Tcw_pred7, const RumiKeyPoint *last_keys_un, int32_t nlast,
                                 const int32_t *last_mp, const uint8_t *last_outlier, const RumiTrackPoints *pts, float th_motion, int32_t *frame_mp,
                                 int32_t *discarded, RumiTrackResult *res) {
    if (!t || !K4 || !Tcw_pred7 || !res || !frame_mp || !discarded || nlast < 0 || (nlast > 0 && (!last_keys_un || !last_mp || !last_outlier)) ||
        track_check_points(pts, false) != RUMI_OK)
        return RUMI_E_INVALID;
    if (t->curN < 0) { g_lastError = "rumi_track_motion: no frame is resident (rumi_track_extract first)"; return RUMI_E_INVALID; }
    RumiMatcher *m = t->m;
    const int n = t->curN, nmp = pts->n;
    if (nlast > m->maxQ || nmp > t->maxPts) { g_lastError = "rumi_track_motion: more points / features than the tracker was created for"; return RUMI_E_CAPACITY; }
    for (int i = 0; i < nlast; i++) if (last_mp[i] >= nmp) { g_lastError = "rumi_track_motion: last_mp index outside the point table"; return RUMI_E_INVALID; }
    HIP_TRY(hipSetDevice(t->device));
    std::memset(res, 0, sizeof(*res));
    t->projN = 0;
    res->n = n; res->mono_index = t->curMono; res->th_motion = (int32_t)th_motion;
    std::memcpy(res->Tcw_motion, Tcw_pred7, 28); std::memcpy(res->Tcw, Tcw_pred7, 28);
    for (int i = 0; i < n; i++) { frame_mp[i] = -1; discarded[i] = -1; }
    TrackBlock *dB = reinterpret_cast<TrackBlock *>(t->dBlk);
    int32_t *dMpOut = reinterpret_cast<int32_t *>(t->dBlk + t->oMp);
    uint8_t *dOutF = t->dBlk + t->oOut;
    FrameDev fd;
    int rc = track_frame_dev(t, &fd);
    if (rc != RUMI_OK) return rc;
    float pose[11];
    std::memcpy(pose, Tcw_pred7, 7 * sizeof(float)); std::memcpy(pose + 7, K4, 4 * sizeof(float));
    H2D(m->dPose, pose, 11);
    if (nmp > 0) { H2D(m->dF[0], pts->pos, (size_t)nmp * 3); H2D(m->dI[1], pts->obs, nmp); H2D(m->dQDesc, pts->desc, (size_t)nmp * 32); }
    if (nlast > 0) { H2D(m->dQKeys, last_keys_un, nlast); H2D(m->dI[0], last_mp, nlast); H2D(m->dU8a, last_outlier, nlast); }
    FLUSH(m);
    const int gI = std::max(1, (std::max(std::max(n, nmp), 4) + 255) / 256), gC = std::max(1, (t->cap + 255) / 256);
    hipLaunchKernelGGL(k_track_init, dim3(gI), dim3(256), 0, nullptr, n, nmp, 1, m->dFeatMp, m->dOut, t->dSeen, dOutF, dMpOut, m->dPose, dB);
    const bool small = n <= kTrackLdsEdges;
    // the usual case (>= 20 matches at th, no list overflow) in one queue, as in rumi_track_frame: the search's result header and the map-point
    // vector it leaves travel back with the results; anything else is redone stage by stage below
    static const int envSpec = std::getenv("RUMI_TRACK_SPECULATE") ? std::atoi(std::getenv("RUMI_TRACK_SPECULATE")) : 1;
    static const bool noFusedLists = std::getenv("RUMI_MATCH_NO_FUSED") != nullptr;
    if (envSpec && !noFusedLists && n > 0 && nlast > 0 && nmp > 0 && m->listCap / (size_t)nlast >= 64) {
        int32_t *dSnap = reinterpret_cast<int32_t *>(t->dBlk + t->oMpM);
        hipLaunchKernelGGL(k_queries_frame, dim3((nlast + 255) / 256), dim3(256), 0, nullptr, nlast, m->dQKeys, m->dI[0], m->dU8a, m->dF[0], m->dI[1], m->dPose,
                           m->dPose + 7, m->dScale, th_motion, fd.minX, fd.minY, fd.maxX, fd.maxY, m->dQ);
        if ((rc = build_lists(m, MODE_FRAME, nlast, fd, m->dQDesc, false, true)) != RUMI_OK) return rc;
        ResolveArgs A{MODE_FRAME, nlast, fd.n, m->dQ, m->dCounts, m->dOffsets, m->dLists, fd.keys, m->dI[1], m->dFeatMp, m->dAssign, m->dNmatches,
                      0.f, 1, nullptr, 0.f, 0, m->dOverflow, m->dF[0], t->dInvSigma2, t->dXw, t->dObs, t->dW, t->dIdx, dB->start, dSnap};   // (ends with the gather)
        hipLaunchKernelGGL(k_resolve, dim3(1), dim3(1024), (size_t)std::max(fd.n, 1) * sizeof(int32_t), nullptr, A);
        if ((rc = rumi::pose_opt_device(dB->start, t->dXw, t->dObs, t->dW, m->dPose + 7, m->dPose, dB->Tout, t->dOutC, dB->nGood, t->dActive, t->dChi, small, nullptr)) != RUMI_OK) return rc;
        hipLaunchKernelGGL(k_track_discard, dim3(gC), dim3(256), 0, nullptr, t->dIdx, t->dOutC, m->dFeatMp, m->dI[1], dB, (const int32_t *)m->dOut);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(t->hBlk + t->oMp, m->dFeatMp, (size_t)n * 4, hipMemcpyDeviceToHost, nullptr));
        HIP_TRY(hipMemcpyAsync(t->hBlk + t->oMpM, dSnap, (size_t)n * 4, hipMemcpyDeviceToHost, nullptr));
        HIP_TRY(hipMemcpy(t->hBlk, t->dBlk, sizeof(TrackBlock), hipMemcpyDeviceToHost));
        const TrackBlock *hS = reinterpret_cast<const TrackBlock *>(t->hBlk);
        if (hS->spec[0] >= 20 && hS->spec[1] == 0) {
            res->nmatches_motion = hS->spec[0];
            std::memcpy(res->Tcw_motion, hS->Tout, 28); std::memcpy(res->Tcw, hS->Tout, 28);
            res->ngood_motion = hS->nGood[0]; res->nmatches_map = hS->counters[0];
            const int32_t *before = reinterpret_cast<const int32_t *>(t->hBlk + t->oMpM), *after = reinterpret_cast<const int32_t *>(t->hBlk + t->oMp);
            for (int i = 0; i < n; i++) { if (before[i] >= 0 && after[i] < 0) discarded[i] = before[i]; frame_mp[i] = after[i]; }
            return RUMI_OK;
        }
        hipLaunchKernelGGL(k_track_init, dim3(gI), dim3(256), 0, nullptr, n, nmp, 1, m->dFeatMp, m->dOut, t->dSeen, dOutF, dMpOut, m->dPose, dB);
    }
    int nm = 0;
    std::vector<int32_t> searched((size_t)std::max(n, 1), -1);
    for (int attempt = 0; attempt < 2 && n > 0 && nlast > 0 && nmp > 0; attempt++) {      // Tracking.cc:2466-2474
        const float th = attempt == 0 ? th_motion : 2 * th_motion;
        if (attempt == 1) hipLaunchKernelGGL(k_track_init, dim3(gI), dim3(256), 0, nullptr, n, nmp, 0, m->dFeatMp, m->dOut, t->dSeen, dOutF, dMpOut, m->dPose, dB);
        hipLaunchKernelGGL(k_queries_frame, dim3((nlast + 255) / 256), dim3(256), 0, nullptr, nlast, m->dQKeys, m->dI[0], m->dU8a, m->dF[0], m->dI[1], m->dPose,
                           m->dPose + 7, m->dScale, th, fd.minX, fd.minY, fd.maxX, fd.maxY, m->dQ);
        if ((rc = run_search(m, MODE_FRAME, nlast, fd, m->dQDesc, m->dI[1], 0.f, 1, searched.data(), &nm)) != RUMI_OK) return rc;
        res->th_motion = (int32_t)th;
        if (nm >= 20) break;
    }
    res->nmatches_motion = nm;
    if (n > 0) std::memcpy(frame_mp, searched.data(), (size_t)n * 4);
    if (nm < 20) return RUMI_OK;                                  // TrackWithMotionModel returns false here (:2476-2483): nothing else has happened to the frame
    hipLaunchKernelGGL(k_track_gather, dim3(1), dim3(1024), 0, nullptr, n, fd.keys, m->dFeatMp, m->dF[0], t->dInvSigma2, t->dXw, t->dObs, t->dW, t->dIdx, dB->start);
    if ((rc = rumi::pose_opt_device(dB->start, t->dXw, t->dObs, t->dW, m->dPose + 7, m->dPose, dB->Tout, t->dOutC, dB->nGood, t->dActive, t->dChi, small, nullptr)) != RUMI_OK) return rc;
    hipLaunchKernelGGL(k_track_discard, dim3(gC), dim3(256), 0, nullptr, t->dIdx, t->dOutC, m->dFeatMp, m->dI[1], dB);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(t->hBlk + t->oMp, m->dFeatMp, (size_t)n * 4, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipMemcpy(t->hBlk, t->dBlk, sizeof(TrackBlock), hipMemcpyDeviceToHost));
    const TrackBlock *hB = reinterpret_cast<const TrackBlock *>(t->hBlk);
    std::memcpy(res->Tcw_motion, hB->Tout, 28); std::memcpy(res->Tcw, hB->Tout, 28);
    res->ngood_motion = hB->nGood[0]; res->nmatches_map = hB->counters[0];
    const int32_t *after = reinterpret_cast<const int32_t *>(t->hBlk + t->oMp);
    for (int i = 0; i < n; i++) { if (searched[i] >= 0 && after[i] < 0) discarded[i] = searched[i]; frame_mp[i] = after[i]; }
    return RUMI_OK;
}

extern "C" int rumi_track_reference_keyframe(RumiTracker *t, RumiVocabulary *voc, int32_t levelsup, const float *K4, const float *Tcw_init7,
                                             const RumiFrameFeatures *KF, const RumiFeatureVector *kf_fv, const int32_t *kf_mp,
                                             const RumiTrackPoints *pts, float nnratio, int32_t check_orientation, uint32_t *word_id, double *word_weight,
                                             uint32_t *node_id, int32_t *frame_mp, int32_t *discarded, RumiTrackResult *res) {
    if (!t || !voc || !K4 || !Tcw_init7 || !KF || !kf_fv || !res || !frame_mp || !discarded || !word_id || !word_weight || !node_id || KF->n < 0 ||
        kf_fv->n_nodes < 0 || (KF->n > 0 && (!kf_mp || !KF->keys_un || !KF->desc)) || track_check_points(pts, false) != RUMI_OK)
        return RUMI_E_INVALID;
    if (t->curN < 0) { g_lastError = "rumi_track_reference_keyframe: no frame is resident (rumi_track_extract first)"; return RUMI_E_INVALID; }
    RumiMatcher *m = t->m;
    const int n = t->curN, nmp = pts->n;
    const int nqe = kf_fv->n_nodes > 0 ? kf_fv->offsets[kf_fv->n_nodes] : 0;
    if (KF->n > m->maxQ || nqe > m->maxQ || nmp > t->maxPts || nmp > m->maxQ || kf_fv->n_nodes > m->maxQ) {
        g_lastError = "rumi_track_reference_keyframe: sizes exceed the tracker's capacities"; return RUMI_E_CAPACITY;
    }
    for (int i = 0; i < KF->n; i++) if (kf_mp[i] >= nmp) { g_lastError = "rumi_track_reference_keyframe: kf_mp index outside the point table"; return RUMI_E_INVALID; }
    HIP_TRY(hipSetDevice(t->device));
    std::memset(res, 0, sizeof(*res));
    t->projN = 0;
    res->n = n; res->mono_index = t->curMono;
    std::memcpy(res->Tcw_motion, Tcw_init7, 28); std::memcpy(res->Tcw, Tcw_init7, 28);
    for (int i = 0; i < n; i++) { frame_mp[i] = -1; discarded[i] = -1; }
    if (n == 0) return RUMI_OK;
    TrackBlock *dB = reinterpret_cast<TrackBlock *>(t->dBlk);
    int32_t *dMpOut = reinterpret_cast<int32_t *>(t->dBlk + t->oMp);
    uint8_t *dOutF = t->dBlk + t->oOut;
    const uint8_t *dRecord = t->dBlk + t->oRec;
    // ---- Frame::ComputeBoW (Frame.cc:763-768): the tree descent of every descriptor, then the FeatureVector, both on the device
    int rc = rumi_voc_transform_batch_device(voc, dRecord + 8 + (size_t)t->cap * sizeof(RumiKeyPoint), dRecord, 1, t->cap, levelsup, t->dWord, t->dWeight, t->dNode, nullptr);
    if (rc != RUMI_OK) return rc;
    int npad = 1;
    while (npad < n) npad <<= 1;
    const size_t fvLds = (size_t)npad * sizeof(unsigned long long);
    if (fvLds > 64 * 1024) HIP_TRY(raise_lds_limit(reinterpret_cast<const void *>(k_fv_build), fvLds));
    hipLaunchKernelGGL(k_fv_build, dim3(1), dim3(kFvThreads), fvLds, nullptr, n, npad, t->dNode, t->dWeight, m->dNodesB, m->dOffB, m->dFvIdx, t->dNN);
    // ---- SearchByBoW(pKF, F, vpMapPointMatches) (ORBmatcher.cc:198-370): the key-frame side comes from the host, the frame side is resident
    FrameDev fd;
    if ((rc = track_frame_dev(t, &fd)) != RUMI_OK) return rc;
    float pose[11];
    std::memcpy(pose, Tcw_init7, 7 * sizeof(float)); std::memcpy(pose + 7, K4, 4 * sizeof(float));
    H2D(m->dPose, pose, 11);
    if (KF->n > 0) { H2D(m->dQKeys, KF->keys_un, KF->n); H2D(m->dQDesc, KF->desc, (size_t)KF->n * 32); H2D(m->dI[0], kf_mp, KF->n); }
    if (nmp > 0) { H2D(m->dU8a, pts->bad, nmp); H2D(m->dF[0], pts->pos, (size_t)nmp * 3); H2D(m->dI[1], pts->obs, nmp); }
    if (kf_fv->n_nodes > 0) { H2D(m->dNodesA, kf_fv->node_ids, kf_fv->n_nodes); H2D(m->dOffA, kf_fv->offsets, kf_fv->n_nodes + 1); }
    if (nqe > 0) H2D(m->dIdxA, kf_fv->indices, nqe);
    m->gridPending = false;                                 // candidates come from the FeatureVectors: the spatial grid is not read
    FLUSH(m);
    const int gI = std::max(1, (std::max(std::max(n, nmp), 4) + 255) / 256), gC = std::max(1, (t->cap + 255) / 256);
    hipLaunchKernelGGL(k_track_init, dim3(gI), dim3(256), 0, nullptr, n, nmp, 1, m->dFeatMp, m->dOut, t->dSeen, dOutF, dMpOut, m->dPose, dB);
    if (kf_fv->n_nodes > 0)
        hipLaunchKernelGGL(k_queries_bow, dim3((std::max(nqe, 1) + 255) / 256), dim3(256), 0, nullptr, kf_fv->n_nodes, m->dNodesA, m->dOffA,
                           m->dIdxA, m->dI[0], m->dU8a, m->dQKeys, 0, m->dNodesB, m->dOffB, m->dQ, t->dNN);
    int nm = 0;
    std::vector<int32_t> searched((size_t)n, -1);
    if ((rc = run_search(m, MODE_BOW, nqe, fd, m->dQDesc, nullptr, nnratio, check_orientation, searched.data(), &nm)) != RUMI_OK) return rc;
    res->nmatches_motion = nm;
    std::memcpy(frame_mp, searched.data(), (size_t)n * 4);
    // the per-feature transform for the host's mBowVec / mFeatVec (assembled there in feature order: rumi_voc_assemble): ONE copy of the block into
    // pinned memory, queued BEHIND the pose chain (three copies into the caller's pageable arrays sat between the search and PoseOptimization)
    const size_t C = (size_t)t->cap;
    auto bow_out = [&]() {
        std::memcpy(word_weight, t->hBow, (size_t)n * 8); std::memcpy(word_id, t->hBow + C * 8, (size_t)n * 4); std::memcpy(node_id, t->hBow + C * 12, (size_t)n * 4);
    };
    if (nm < 15) {                                          // TrackReferenceKeyFrame returns false here (:2335-2338)
        HIP_TRY(hipMemcpy(t->hBow, t->dBow, C * 16, hipMemcpyDeviceToHost));
        bow_out();
        return RUMI_OK;
    }
    const bool small = n <= kTrackLdsEdges;
    hipLaunchKernelGGL(k_track_gather, dim3(1), dim3(1024), 0, nullptr, n, fd.keys, m->dFeatMp, m->dF[0], t->dInvSigma2, t->dXw, t->dObs, t->dW, t->dIdx, dB->start);
    if ((rc = rumi::pose_opt_device(dB->start, t->dXw, t->dObs, t->dW, m->dPose + 7, m->dPose, dB->Tout, t->dOutC, dB->nGood, t->dActive, t->dChi, small, nullptr)) != RUMI_OK) return rc;
    hipLaunchKernelGGL(k_track_discard, dim3(gC), dim3(256), 0, nullptr, t->dIdx, t->dOutC, m->dFeatMp, m->dI[1], dB);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(t->hBow, t->dBow, C * 16, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipMemcpyAsync(t->hBlk + t->oMp, m->dFeatMp, (size_t)n * 4, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipMemcpy(t->hBlk, t->dBlk, sizeof(TrackBlock), hipMemcpyDeviceToHost));
    bow_out();
    const TrackBlock *hB = reinterpret_cast<const TrackBlock *>(t->hBlk);
    std::memcpy(res->Tcw_motion, hB->Tout, 28); std::memcpy(res->Tcw, hB->Tout, 28);
    res->ngood_motion = hB->nGood[0]; res->nmatches_map = hB->counters[0];
    const int32_t *after = reinterpret_cast<const int32_t *>(t->hBlk + t->oMp);
    for (int i = 0; i < n; i++) { if (searched[i] >= 0 && after[i] < 0) discarded[i] = searched[i]; frame_mp[i] = after[i]; }
    return RUMI_OK;
}

extern "C" int rumi_track_local(RumiTracker *t, const float *K4, const float *Tcw7, const int32_t *frame_mp_in, const RumiTrackPoints *pts,
                                const uint8_t *seen_in, float th_local, int32_t far_points, float th_far_points, int32_t *frame_mp, uint8_t *outlier,
                                uint8_t *in_view, RumiTrackResult *res) {
    if (!t || !K4 || !Tcw7 || !res || !frame_mp || !outlier || !frame_mp_in || track_check_points(pts, true) != RUMI_OK || (pts->n > 0 && !in_view))
        return RUMI_E_INVALID;
    if (t->curN < 0) { g_lastError = "rumi_track_local: no frame is resident (rumi_track_extract first)"; return RUMI_E_INVALID; }
    RumiMatcher *m = t->m;
    const int n = t->curN, nmp = pts->n;
    if (nmp > t->maxPts) { g_lastError = "rumi_track_local: more points than the tracker was created for"; return RUMI_E_CAPACITY; }
    for (int i = 0; i < n; i++) if (frame_mp_in[i] >= nmp) { g_lastError = "rumi_track_local: frame_mp_in index outside the point table"; return RUMI_E_INVALID; }
    HIP_TRY(hipSetDevice(t->device));
    std::memset(res, 0, sizeof(*res));
    t->projN = 0;
    res->n = n; res->mono_index = t->curMono;
    std::memcpy(res->Tcw_motion, Tcw7, 28); std::memcpy(res->Tcw, Tcw7, 28);
    // SearchLocalPoints, first loop (Tracking.cc:2998-3010), on the host while the arrays are being staged: a bad point leaves the frame, the
    // others are "seen in this frame"; seen_in carries the points the caller's discard loop has marked (mnLastFrameSeen == mCurrentFrame.mnId)
    std::vector<int32_t> mpIn((size_t)std::max(n, 1), -1);
    std::vector<uint8_t> seen((size_t)std::max(nmp, 1), 0);
    if (seen_in) for (int j = 0; j < nmp; j++) seen[j] = seen_in[j] ? 2 : 0;       // the caller's discard loop: 2 (k_track_frustum tells them from the frame's own points)
    for (int i = 0; i < n; i++) {
        const int mp = frame_mp_in[i];
        if (mp < 0) continue;
        if (pts->bad[mp]) continue;
        mpIn[i] = mp; seen[mp] = 1;
    }
    TrackBlock *dB = reinterpret_cast<TrackBlock *>(t->dBlk);
    int32_t *dMpOut = reinterpret_cast<int32_t *>(t->dBlk + t->oMp), *dMpMotion = reinterpret_cast<int32_t *>(t->dBlk + t->oMpM);
    uint8_t *dOutF = t->dBlk + t->oOut, *dView = t->dBlk + t->oView;
    FrameDev fd;
    int rc = track_frame_dev(t, &fd);
    if (rc != RUMI_OK) return rc;
    float pose[11];
    std::memcpy(pose, Tcw7, 7 * sizeof(float)); std::memcpy(pose + 7, K4, 4 * sizeof(float));
    H2D(m->dPose, pose, 11);
    if (n > 0) H2D(m->dFeatMp, mpIn.data(), n);
    if (nmp > 0) {
        H2D(m->dF[0], pts->pos, (size_t)nmp * 3); H2D(m->dF[1], pts->normal, (size_t)nmp * 3); H2D(m->dF[2], pts->min_dist, nmp); H2D(m->dF[3], pts->max_dist, nmp);
        H2D(m->dI[1], pts->obs, nmp); H2D(m->dQDesc, pts->desc, (size_t)nmp * 32); H2D(t->dBad, pts->bad, nmp); H2D(t->dLocal, pts->local, nmp);
        if (pts->stale_in_view && pts->stale_proj) { H2D(t->dStaleIn, pts->stale_in_view, nmp); H2D(t->dStaleProj, pts->stale_proj, (size_t)nmp * 5); }
        H2D(t->dSeen, seen.data(), nmp);
    }
    FLUSH(m);
    const int gI = std::max(1, (std::max(std::max(n, nmp), 4) + 255) / 256), gC = std::max(1, (t->cap + 255) / 256);
    hipLaunchKernelGGL(k_track_local_init, dim3(gI), dim3(256), 0, nullptr, n, m->dPose, m->dPose + 7, dOutF, dMpOut, m->dOut, dB);
    int nmLocal = 0;
    bool speculate = false;
    std::vector<int32_t> tmpMp((size_t)std::max(n, 1));
    if (nmp > 0 && n > 0) {
        const size_t n16 = ((size_t)nmp + 15) & ~(size_t)15;
        if (n16 * 21 > m->stageCap) { g_lastError = "rumi_track_local: point table exceeds the staging block"; return RUMI_E_CAPACITY; }
        uint8_t *dSkip = m->dU8b;
        float *dX = reinterpret_cast<float *>(m->dStage + n16), *dY = dX + n16, *dC = dY + n16, *dD = dC + n16;
        int32_t *dL = reinterpret_cast<int32_t *>(dD + n16);
        const float logSf = std::log(t->cfg.scale_factor);
        { const FrustumArgs FA{nmp, n, m->dFeatMp, dMpMotion, t->dLocal, t->dSeen, t->dBad, dSkip, m->dOut, dB->pose19, fd.minX,
                           fd.minY, fd.maxX, fd.maxY, logSf, t->nlevels, 0.5f, m->dF[0], m->dF[1], m->dF[2], m->dF[3], dView, dX, dY, dL, dC, dD,
                           m->dI[1], m->dScale, th_local, far_points, th_far_points, m->dQ,
                           (pts->stale_in_view && pts->stale_proj) ? t->dStaleIn : nullptr, (pts->stale_in_view && pts->stale_proj) ? t->dStaleProj : nullptr};
          hipLaunchKernelGGL(k_track_frustum, dim3(gI), dim3(256), 0, nullptr, FA); t->projN16 = n16; t->projN = nmp; }
        // the search's counts are not needed before the end: one queue, the result header travels in the block (a list overflow -- the resolve
        // did not run then, the frame's vector is untouched -- sends the stage through the sizing path)
        static const int envSpec = std::getenv("RUMI_TRACK_SPECULATE") ? std::atoi(std::getenv("RUMI_TRACK_SPECULATE")) : 1;
        static const bool noFusedLists = std::getenv("RUMI_MATCH_NO_FUSED") != nullptr;
        speculate = envSpec && !noFusedLists && m->listCap / (size_t)nmp >= 64;
        if (speculate) {
            if ((rc = build_lists(m, MODE_MAPPOINTS, nmp, fd, m->dQDesc, false, true)) != RUMI_OK) return rc;
            ResolveArgs A{MODE_MAPPOINTS, nmp, fd.n, m->dQ, m->dCounts, m->dOffsets, m->dLists, fd.keys, m->dI[1], m->dFeatMp, m->dAssign, m->dNmatches,
                          0.8f, 0, nullptr, 0.f, 0, m->dOverflow};
            hipLaunchKernelGGL(k_resolve, dim3(1), dim3(1024), (size_t)std::max(fd.n, 1) * sizeof(int32_t), nullptr, A);
        } else if ((rc = run_search(m, MODE_MAPPOINTS, nmp, fd, m->dQDesc, m->dI[1], 0.8f, 0, tmpMp.data(), &nmLocal)) != RUMI_OK) return rc;
    }
    res->nmatches_local = nmLocal;
    const bool small = n <= kTrackLdsEdges;
    const TrackBlock *hB = reinterpret_cast<const TrackBlock *>(t->hBlk);
    for (int pass = 0; pass < 2; pass++) {
        hipLaunchKernelGGL(k_track_gather, dim3(1), dim3(1024), 0, nullptr, n, fd.keys, m->dFeatMp, m->dF[0], t->dInvSigma2, t->dXw, t->dObs, t->dW, t->dIdx, dB->start);
        if ((rc = rumi::pose_opt_device(dB->start, t->dXw, t->dObs, t->dW, m->dPose + 7, dB->Tout, dB->Tout + 7, t->dOutC, dB->nGood + 1, t->dActive, t->dChi, small, nullptr)) != RUMI_OK) return rc;
        hipLaunchKernelGGL(k_track_finish, dim3(gC), dim3(256), 0, nullptr, t->dIdx, t->dOutC, m->dFeatMp, m->dI[1], dOutF, dMpOut, 1, dB, speculate ? (const int32_t *)m->dOut : (const int32_t *)nullptr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(t->hBlk, t->dBlk, t->oRec, hipMemcpyDeviceToHost));       // header, mvpMapPoints, mvbOutlier, mbTrackInView
        if (!speculate) break;
        if (hB->spec[3] == 0) { res->nmatches_local = hB->spec[2]; break; }
        // a candidate list overflowed: the search again with exact list sizes, then the optimisation on its result
        speculate = false;
        hipLaunchKernelGGL(k_track_local_init, dim3(gI), dim3(256), 0, nullptr, n, m->dPose, m->dPose + 7, dOutF, dMpOut, m->dOut, dB);
        if ((rc = run_search(m, MODE_MAPPOINTS, nmp, fd, m->dQDesc, m->dI[1], 0.8f, 0, tmpMp.data(), &nmLocal)) != RUMI_OK) return rc;
        res->nmatches_local = nmLocal;
    }
    std::memcpy(res->Tcw, hB->Tout + 7, 28);
    std::memcpy(res->Rcw, hB->pose19, 36); std::memcpy(res->tcw, hB->pose19 + 9, 12); std::memcpy(res->Ow, hB->pose19 + 12, 12);
    res->ngood_local = hB->nGood[1]; res->matches_inliers = hB->counters[1];
    if (n > 0) { std::memcpy(frame_mp, t->hBlk + t->oMp, (size_t)n * 4); std::memcpy(outlier, t->hBlk + t->oOut, (size_t)n); }
    int nTo = 0;
    if (nmp > 0) {
        if (n > 0) std::memcpy(in_view, t->hBlk + t->oView, (size_t)nmp); else std::memset(in_view, 0, (size_t)nmp);
        for (int j = 0; j < nmp; j++) nTo += in_view[j] == 1;         // (2: a stale flag of an earlier frame, not an isInFrustum of this one)
    }
    res->n_to_match = nTo;
    return RUMI_OK;
}

/* mTrackProjX, mTrackProjY, mnTrackScaleLevel, mTrackViewCos, mTrackDepth of every table point as the SearchLocalPoints of the LAST rumi_track_frame /
 * rumi_track_local call left them (Frame::isInFrustum writes them into the MapPoint, Frame.cc:558-630; the values of a point that is not in view are
 * not meaningful).  They are still in the matcher's staging block: one more copy brings them.  Valid until the next rumi_track_* call. */
extern "C" int rumi_track_last_projections(RumiTracker *t, int32_t n_points, float *proj5_out) {
    if (!t || !proj5_out || n_points < 0) return RUMI_E_INVALID;
    if (t->projN <= 0 || n_points != t->projN) { g_lastError = "rumi_track_last_projections: no SearchLocalPoints result of that size is resident"; return RUMI_E_INVALID; }
    HIP_TRY(hipSetDevice(t->device));
    RumiMatcher *m = t->m;
    const size_t n16 = t->projN16;
    std::vector<float> h(5 * n16);
    HIP_TRY(hipMemcpy(h.data(), m->dStage + n16, 5 * n16 * sizeof(float), hipMemcpyDeviceToHost));
    const float *X = h.data(), *Y = X + n16, *Cc = Y + n16, *D = Cc + n16;
    const int32_t *L = reinterpret_cast<const int32_t *>(D + n16);
    for (int i = 0; i < n_points; i++) { float *o = proj5_out + (size_t)i * 5; o[0] = X[i]; o[1] = Y[i]; o[2] = (float)L[i]; o[3] = Cc[i]; o[4] = D[i]; }
    return RUMI_OK;
}

/* Lens distortion of the camera (Frame::UndistortKeyPoints / ComputeImageBounds, R/lib_src/Frame.cc:770-826; R/config/euroc_ori.yaml:23-31 has
 * k1 = -0.283): K4 = fx, fy, cx, cy of mK, dist5 = mDistCoef (k1, k2, p1, p2, k3).  From the next rumi_track_extract / rumi_track_frame on the
 * resident frame carries mvKeysUn (the grid, every search and PoseOptimization read those) and the undistorted image bounds; keys_out of those calls
 * stays mvKeys, as ExtractORB returns them.  dist5 == NULL or dist5[0] == 0: none (mvKeysUn = mvKeys, the reference's own test, Frame.cc:771). */
extern "C" int rumi_track_set_distortion(RumiTracker *t, const float *K4, const float *dist5) {
    if (!t) return RUMI_E_INVALID;
    if (!dist5 || dist5[0] == 0.0f) { t->distort = false; return RUMI_OK; }
    if (!K4 || !(K4[0] != 0.0f) || !(K4[1] != 0.0f)) { g_lastError = "rumi_track_set_distortion: camera matrix"; return RUMI_E_INVALID; }
    HIP_TRY(hipSetDevice(t->device));
    if (!t->dKeysUn) HIP_TRY(hipMalloc((void **)&t->dKeysUn, (size_t)t->cap * sizeof(RumiKeyPoint)));
    UndistortArgs &A = t->ua;
    A.fx = K4[0]; A.fy = K4[1]; A.cx = K4[2]; A.cy = K4[3]; A.ifx = 1. / A.fx; A.ify = 1. / A.fy;
    A.k1 = dist5[0]; A.k2 = dist5[1]; A.p1 = dist5[2]; A.p2 = dist5[3]; A.k3 = dist5[4];
    t->distort = true;
    t->curN = -1;                                          // a frame extracted under other coefficients is not this camera's
    return RUMI_OK;
}
/* mvKeysUn of the resident frame (keys_un_out [cap >= n]) and {mnMinX, mnMinY, mnMaxX, mnMaxY} (bounds4); either may be NULL. */
extern "C" int rumi_track_undistorted(RumiTracker *t, RumiKeyPoint *keys_un_out, int32_t cap, float *bounds4) {
    if (!t) return RUMI_E_INVALID;
    if (t->curN < 0) { g_lastError = "rumi_track_undistorted: no frame is resident"; return RUMI_E_INVALID; }
    if (bounds4) std::memcpy(bounds4, t->bounds, 16);
    if (keys_un_out && t->curN > 0) {
        if (cap < t->curN) return RUMI_E_CAPACITY;
        HIP_TRY(hipSetDevice(t->device));
        const RumiKeyPoint *src = t->distort ? t->dKeysUn : reinterpret_cast<const RumiKeyPoint *>(t->dBlk + t->oRec + 8);
        HIP_TRY(hipMemcpy(keys_un_out, src, (size_t)t->curN * sizeof(RumiKeyPoint), hipMemcpyDeviceToHost));
    }
    return RUMI_OK;
}
