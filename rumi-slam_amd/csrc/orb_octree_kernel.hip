// On-device quadtree key-point distribution + output slot assignment (gfx950).
//
//   k_octree    one 256-thread workgroup per (frame, level): DistributeOctTree, R/lib_src/ORBextractor.cc:538-724.
//               Keys stay where the FAST kernel left them (cand[], HBM/L2); each key only carries a 16-bit owner
//               id (owner[], HBM/L2).  The node pool, the list links, the open/sort arrays live in LDS.
//               Lane-parallel sweeps relabel keys and count quadrant populations with LDS atomics; thread 0 runs
//               the list choreography + the replayed std::sort (orb_octree.h) between barriers, so the result
//               ORDER equals the reference's.
//   k_assemble  one workgroup per frame: concatenates the levels and assigns the output slot of every key-point
//               by the lapping-area rule of operator() (:1067-1088) with a block scan (the reference walks them
//               serially with monoIndex++ / stereoIndex--).
#include <hip/hip_runtime.h>

#include "orb_device.h"
#include "orb_octree.h"

namespace rumi {

constexpr int kMaxRoots = 16;

__host__ __device__ inline int octree_pool_cap(int N, int nIni) { return 2 * (N > nIni ? N : nIni) + 16 + nIni; }
// LDS bytes of one k_octree workgroup for a pool of `cap` nodes
__host__ __device__ inline size_t octree_lds_bytes(int cap) {
    return (size_t)cap * (sizeof(OctNode) + 2 * sizeof(OctEntry) + 2 * sizeof(uint16_t)) + 64;
}

__global__ __launch_bounds__(256) void k_octree(const DevParams *__restrict__ P, const uint32_t *__restrict__ cand,
                                                const int32_t *__restrict__ levelStart, uint16_t *__restrict__ owner,
                                                uint32_t *__restrict__ selLevel, int32_t *__restrict__ selLevelCnt,
                                                int selLevelCap, int32_t *__restrict__ errFlag) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ int sPhase, sSize, sRootId[kMaxRoots];
    __shared__ unsigned int sRootN[kMaxRoots];

    const int tid = threadIdx.x, level = blockIdx.x, frame = blockIdx.y;
    const DevLevel &L = P->lv[level];
    const int32_t *ls = levelStart + (long long)frame * (kMaxLevels + 1);
    const int n = ls[level + 1] - ls[level];
    const uint32_t *c = cand + (long long)frame * P->totalCand + ls[level];
    uint16_t *own = owner + (long long)frame * P->totalCand + ls[level];
    uint32_t *out = selLevel + ((long long)frame * P->nlevels + level) * selLevelCap;
    int32_t *outCnt = selLevelCnt + (long long)frame * P->nlevels + level;
    if (n <= 0) {
        if (tid == 0) *outCnt = 0;
        return;
    }
    const int N = L.nfeat;
    const int W = L.maxBX - kBorder, Hh = L.maxBY - kBorder;
    const int nIni = (int)__builtin_roundf((float)W / (float)Hh);
    if (nIni <= 0 || nIni > kMaxRoots) {           // the reference divides by zero / we do not stage that many roots
        if (tid == 0) { *outCnt = 0; atomicOr(errFlag, 1); }
        return;
    }
    const float hX = (float)W / nIni;
    const int cap = octree_pool_cap(N, nIni);

    OctNode *nodes = reinterpret_cast<OctNode *>(lds);
    OctEntry *open = reinterpret_cast<OctEntry *>(nodes + cap);
    OctEntry *prev = open + cap;
    uint16_t *freeIds = reinterpret_cast<uint16_t *>(prev + cap);
    uint16_t *splitIds = freeIds + cap;

    OctState s{nodes, freeIds, open, prev, splitIds, cap, 0, 0, 0, kNil, 0, N, 0, 0, 0};
    if (tid < kMaxRoots) sRootN[tid] = 0;
    if (tid == 0) {
        for (int i = cap - 1; i >= 0; i--) s.freeIds[s.nFree++] = (uint16_t)i;
        int tail = kNil;
        for (int i = 0; i < nIni; i++) {          // :548-561  roots in push_back order
            const int id = oct_alloc(s);
            OctNode &r = s.nodes[id];
            r.x0 = (uint16_t)(int)(hX * (float)i); r.x1 = (uint16_t)(int)(hX * (float)(i + 1));
            r.y0 = 0; r.y1 = (uint16_t)Hh;
            r.next = kNil; r.prev = (uint16_t)tail;
            r.n = 0; r.noMore = 0; r.split = 0;
            r.cnt[0] = r.cnt[1] = r.cnt[2] = r.cnt[3] = 0;
            r.child[0] = r.child[1] = r.child[2] = r.child[3] = kNil;
            if (tail != kNil) s.nodes[tail].next = (uint16_t)id; else s.head = id;
            tail = id; s.size++;
            sRootId[i] = id;
        }
    }
    __syncthreads();
    // :564-567  keys -> roots
    for (int i = tid; i < n; i += 256) {
        const int r = (int)((float)cand_x(c[i]) / hX);
        own[i] = (uint16_t)sRootId[r];
        atomicAdd(&sRootN[r], 1u);
    }
    __syncthreads();
    if (tid == 0) {                                // :570-578
        for (int i = 0; i < nIni; i++) {
            OctNode &r = s.nodes[sRootId[i]];
            r.n = (uint16_t)sRootN[i];
            if (r.n == 1) r.noMore = 1;
            else if (r.n == 0) { oct_erase(s, sRootId[i]); s.freeIds[s.nFree++] = (uint16_t)sRootId[i]; }
        }
    }
    __syncthreads();
    // quadrant populations of the roots
    for (int i = tid; i < n; i += 256) {
        OctNode &nd = nodes[own[i]];
        if (!nd.noMore) {
            const int q = oct_quadrant(nd, cand_x(c[i]), cand_y(c[i]));
            atomicAdd(reinterpret_cast<unsigned int *>(&nd.cnt[q & 2]), 1u << (16 * (q & 1)));
        }
    }
    __syncthreads();

    // while (!bFinish)  :587-702
    while (true) {
        if (tid == 0) {
            oct_round(s);
            sPhase = s.phase;
        }
        __syncthreads();
        for (int i = tid; i < n; i += 256) {       // relabel keys of divided nodes, count inside the new owners
            int id = own[i];
            if (!nodes[id].split) continue;
            const int x = cand_x(c[i]), y = cand_y(c[i]);
            id = nodes[id].child[oct_quadrant(nodes[id], x, y)];
            own[i] = (uint16_t)id;
            OctNode &nd = nodes[id];
            if (!nd.noMore) {
                const int q = oct_quadrant(nd, x, y);
                atomicAdd(reinterpret_cast<unsigned int *>(&nd.cnt[q & 2]), 1u << (16 * (q & 1)));
            }
        }
        __syncthreads();
        if (tid == 0) oct_release_split(s);
        if (sPhase == 2) break;
        __syncthreads();
    }
    // :705-721  best key of every node, nodes in list order.  `prev` is free now: reuse it as the order array.
    uint16_t *order = reinterpret_cast<uint16_t *>(prev);
    if (tid == 0) {
        int r = 0;
        for (int it = s.head; it != kNil; it = nodes[it].next) { order[r++] = (uint16_t)it; nodes[it].best = 0; }
        sSize = r;
        if (s.overflow) atomicOr(errFlag, 2);
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256)
        atomicMax(&nodes[own[i]].best, ((uint32_t)cand_score(c[i]) << 16) | (uint32_t)(0xFFFF - i));
    __syncthreads();
    const int m = sSize;
    if (m > selLevelCap) {
        if (tid == 0) { *outCnt = 0; atomicOr(errFlag, 4); }
        return;
    }
    for (int r = tid; r < m; r += 256) out[r] = c[0xFFFF - (int)(nodes[order[r]].best & 0xFFFF)];
    if (tid == 0) *outCnt = m;
}

// Concatenate levels, assign slots: in (level, list) order, key-points with lap0 <= x*scale <= lap1 fill the
// output from the back (stereoIndex--), the others from the front (monoIndex++).
__global__ __launch_bounds__(256) void k_assemble(const DevParams *__restrict__ P, const uint32_t *__restrict__ selLevel,
                                                  const int32_t *__restrict__ selLevelCnt, int selLevelCap, int lap0,
                                                  int lap1, uint32_t *__restrict__ selPacked, uint32_t *__restrict__ selMeta,
                                                  int32_t *__restrict__ selCount, int selCap, int32_t *__restrict__ counts,
                                                  int32_t *__restrict__ errFlag) {
    __shared__ int lvStart[kMaxLevels + 1];
    __shared__ int part[256];
    const int tid = threadIdx.x, frame = blockIdx.x;
    const int nl = P->nlevels;
    if (tid == 0) {
        int run = 0;
        for (int l = 0; l < nl; l++) { lvStart[l] = run; run += selLevelCnt[(long long)frame * nl + l]; }
        lvStart[nl] = run;
    }
    __syncthreads();
    const int total = lvStart[nl];
    if (total > selCap) {
        if (tid == 0) { selCount[frame] = 0; counts[2 * frame] = total; counts[2 * frame + 1] = 0; atomicOr(errFlag, 8); }
        return;
    }
    const int chunk = (total + 255) / 256;
    const int k0 = tid * chunk, k1 = min(total, k0 + chunk);
    // pass 1: flags of my contiguous chunk
    int level = 0, nflag = 0;
    for (int k = k0; k < k1; k++) {
        while (k >= lvStart[level + 1]) level++;
        const uint32_t pk = selLevel[((long long)frame * nl + level) * selLevelCap + (k - lvStart[level])];
        float x = (float)((int)(pk & 0xFFF) + kBorder);
        if (level != 0) x = x * P->lv[level].scale;
        nflag += (x >= (float)lap0 && x <= (float)lap1) ? 1 : 0;
    }
    part[tid] = nflag;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < 256; i++) { const int t = part[i]; part[i] = run; run += t; }
        selCount[frame] = total;
        counts[2 * frame] = total;
        counts[2 * frame + 1] = total - run;      // monoIndex
    }
    __syncthreads();
    int before = part[tid];                       // flagged key-points before k0
    level = 0;
    for (int k = k0; k < k1; k++) {
        while (k >= lvStart[level + 1]) level++;
        const uint32_t pk = selLevel[((long long)frame * nl + level) * selLevelCap + (k - lvStart[level])];
        float x = (float)((int)(pk & 0xFFF) + kBorder);
        if (level != 0) x = x * P->lv[level].scale;
        const bool f = x >= (float)lap0 && x <= (float)lap1;
        const int slot = f ? (total - 1 - before) : (k - before);
        before += f ? 1 : 0;
        selPacked[(long long)frame * selCap + k] = pk;
        selMeta[(long long)frame * selCap + k] = (uint32_t)level | ((uint32_t)slot << 8);
    }
}

void launch_octree(const DevParams *dP, const DevParams &hP, const uint32_t *cand, const int32_t *levelStart,
                   uint16_t *owner, uint32_t *selLevel, int32_t *selLevelCnt, int selLevelCap, int32_t *errFlag,
                   int nframes, size_t ldsBytes, hipStream_t st) {
    static size_t attrSet = 0;
    if (ldsBytes > attrSet) {   // > 64 KiB of dynamic LDS needs the opt-in
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_octree), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
        attrSet = ldsBytes;
    }
    hipLaunchKernelGGL(k_octree, dim3(hP.nlevels, nframes), dim3(256), ldsBytes, st, dP, cand, levelStart, owner, selLevel,
                       selLevelCnt, selLevelCap, errFlag);
}
void launch_assemble(const DevParams *dP, const uint32_t *selLevel, const int32_t *selLevelCnt, int selLevelCap, int lap0,
                     int lap1, uint32_t *selPacked, uint32_t *selMeta, int32_t *selCount, int selCap, int32_t *counts,
                     int32_t *errFlag, int nframes, hipStream_t st) {
    hipLaunchKernelGGL(k_assemble, dim3(nframes), dim3(256), 0, st, dP, selLevel, selLevelCnt, selLevelCap, lap0, lap1,
                       selPacked, selMeta, selCount, selCap, counts, errFlag);
}
size_t octree_lds_for(const DevParams &hP) {
    size_t mx = 0;
    for (int l = 0; l < hP.nlevels; l++) {
        const int W = hP.lv[l].maxBX - kBorder, Hh = hP.lv[l].maxBY - kBorder;
        int nIni = (int)__builtin_roundf((float)W / (float)Hh);
        if (nIni < 1) nIni = 1;
        mx = std::max(mx, octree_lds_bytes(octree_pool_cap(hP.lv[l].nfeat, nIni)));
    }
    return mx;
}

}  // namespace rumi
