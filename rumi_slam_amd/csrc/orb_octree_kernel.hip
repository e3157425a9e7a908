// On-device quadtree key-point distribution + output slot assignment (gfx950).
//
//   k_octree    one 512-thread workgroup per (frame, level): DistributeOctTree, R/lib_src/ORBextractor.cc:538-724.
//               Keys stay where the FAST kernel left them (cand[], HBM/L2); each key only carries a 16-bit owner
//               id (owner[], HBM/L2).  The node pool, the list (an array in list order), the open/sort arrays live in LDS.
//               Lane-parallel sweeps relabel keys and count quadrant populations with LDS atomics; the list passes
//               (divide, push children to the front, erase parents) are workgroup prefix sums + scatters, std::sort is
//               replayed by the whole workgroup (wg_sort_like_libstdcxx), so the result ORDER equals the reference's.
//   k_assemble  one workgroup per frame: concatenates the levels and assigns the output slot of every key-point
//               by the lapping-area rule of operator() (:1067-1088) with a block scan (the reference walks them
//               serially with monoIndex++ / stereoIndex--).
#include <hip/hip_runtime.h>

#include "orb_device.h"
#include "rumi_common.h"
#include "orb_octree.h"

namespace rumi {

constexpr int kMaxRoots = 16;
constexpr int kOctThreads = 512;

__host__ __device__ inline int octree_pool_cap(int N, int nIni) { return 2 * (N > nIni ? N : nIni) + 16 + nIni; }
// LDS bytes of one k_octree workgroup for a pool of `cap` nodes: nodes, open + sort arrays, free stack, split stack, two list arrays
__host__ __device__ inline size_t octree_lds_bytes(int cap) {
    return (size_t)cap * (sizeof(OctNode) + 2 * sizeof(OctEntry) + 4 * sizeof(uint16_t)) + 2 * (size_t)(cap / 16 + 2) * 8 + 64;   // + two SortSeg work lists
}

// exclusive scan of one 64-bit value per thread over the workgroup (three packed 20-bit counters); *total = sum over all threads
__device__ __forceinline__ unsigned long long block_scan64(unsigned long long v, unsigned long long *sWave, unsigned long long *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) sWave[wave] = inc;
    __syncthreads();
    unsigned long long base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kOctThreads / 64; w++) {
        const unsigned long long t = sWave[w];
        if (w < wave) base += t;
        tot += t;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// ---- workgroup-parallel replay of libstdc++'s std::sort on (key, id) entries -------------------------------------------------
// std::sort = introsort loop (median-of-3 pivot moved to the front, Hoare "unguarded" partition, recursion on the right part,
// depth limit 2*floor(log2 n) with a heap-sort fallback) down to segments of <= 16, then one insertion sort over everything.
// The segments of one recursion level are disjoint, so they are partitioned concurrently, one WAVE per segment; inside a
// segment the Hoare partition is data-parallel: the k-th stop of the left pointer (element not < pivot) is swapped with the
// k-th stop of the right pointer (element not > pivot) while the former lies left of the latter, which only needs the ranks
// of the stop positions.  The final insertion sort is stable and never moves an element out of its <= 16-element leaf, so it
// equals a stable rank over a +-15 window.  Tie order of equal keys therefore matches libstdc++ exactly (tests compare with
// the real std::sort).
struct SortSeg { uint16_t first, last; int16_t depth; uint16_t pad; };

__device__ __forceinline__ void wave_fence_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// a[0..n) sorted in place; tmp[n] entries, sf/sr[n] uint16, segA/segB[n/16+2] are scratch; every thread of the workgroup calls it
__device__ void wg_sort_like_libstdcxx(OctEntry *a, int n, OctEntry *tmp, uint16_t *sf, uint16_t *sr, SortSeg *segA, SortSeg *segB,
                                       int *sCount /* [2] in LDS */) {
    using namespace sortimpl;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nWaves = blockDim.x >> 6;
    if (n <= 1) return;
    if (tid == 0) {
        int lg = 0;
        for (int t = n; t > 1; t >>= 1) lg++;
        sCount[0] = 0; sCount[1] = 0;
        if (n > 16) { segA[0] = SortSeg{0, (uint16_t)n, (int16_t)(lg * 2), 0}; sCount[0] = 1; }
    }
    __syncthreads();
    int cur = 0;
    while (true) {
        const int nSeg = sCount[cur];
        if (nSeg == 0) break;
        SortSeg *in = cur ? segB : segA, *outS = cur ? segA : segB;
        for (int si = wave; si < nSeg; si += nWaves) {
            const SortSeg sg = in[si];
            const int first = sg.first, last = sg.last;
            if (sg.depth == 0) {                                        // std::__partial_sort(first, last, last)
                if (lane == 0) heap_sort(a + first, a + last);
                continue;
            }
            if (lane == 0) move_median_to_first(a + first, a + first + 1, a + first + (last - first) / 2, a + last - 1);
            wave_fence_lds();
            const uint32_t piv = a[first].key;
            // stops of the left pointer, in ascending order: sf[first + k]
            int nF = 0, nR = 0;
            for (int base = first + 1; base < last; base += 64) {
                const int i = base + lane;
                const bool stop = i < last && !(a[i].key < piv);
                const unsigned long long b = __ballot(stop);
                if (stop) sf[first + nF + __popcll(b & ((1ull << lane) - 1ull))] = (uint16_t)i;
                nF += __popcll(b);
            }
            // stops of the right pointer, in descending order: sr[first + k]
            for (int base = last - 1; base > first; base -= 64) {
                const int i = base - lane;
                const bool stop = i > first && !(piv < a[i].key);
                const unsigned long long b = __ballot(stop);
                if (stop) sr[first + nR + __popcll(b & ((1ull << lane) - 1ull))] = (uint16_t)i;
                nR += __popcll(b);
            }
            wave_fence_lds();
            int swaps = 0;
            const int nPair = min(nF, nR);
            for (int base = 0; base < nPair; base += 64) {
                const int k = base + lane;
                bool sw = false;
                if (k < nPair) {
                    const int f = sf[first + k], r = sr[first + k];
                    if (f < r) { sw = true; const OctEntry t = a[f]; a[f] = a[r]; a[r] = t; }
                }
                swaps += __popcll(__ballot(sw));
            }
            wave_fence_lds();
            if (lane == 0) {
                int cut;
                if (swaps >= 1) {
                    const int nextF = swaps < nF ? (int)sf[first + swaps] : 0x7FFFFFFF;
                    cut = min(nextF, (int)sr[first + swaps - 1]);
                } else {
                    cut = sf[first];
                }
                const int16_t d = (int16_t)(sg.depth - 1);
                if (last - cut > 16) outS[atomicAdd(&sCount[cur ^ 1], 1)] = SortSeg{(uint16_t)cut, (uint16_t)last, d, 0};
                if (cut - first > 16) outS[atomicAdd(&sCount[cur ^ 1], 1)] = SortSeg{(uint16_t)first, (uint16_t)cut, d, 0};
            }
        }
        __syncthreads();
        if (tid == 0) sCount[cur] = 0;
        cur ^= 1;
        __syncthreads();
    }
    // __final_insertion_sort == stable sort inside each leaf == stable rank over a +-15 window
    for (int i = tid; i < n; i += blockDim.x) {
        const OctEntry e = a[i];
        int pos = i;
        for (int j = max(0, i - 15); j < i; j++) pos -= a[j].key > e.key;
        for (int j = i + 1; j < min(n, i + 16); j++) pos += a[j].key < e.key;
        tmp[pos] = e;
    }
    __syncthreads();
    for (int i = tid; i < n; i += blockDim.x) a[i] = tmp[i];
    __syncthreads();
}

struct OctLds {
    OctNode *nodes;
    OctEntry *open, *prev;
    uint16_t *freeIds, *splitIds, *listA, *listB;
    SortSeg *segA, *segB;
};

// DivideNode + the push_front block after it (ORBextractor.cc:471-522, :603-637) for ONE parent, given where its children land:
// g0 = creation index of its first child in this round (the list receives children in REVERSE creation order, because every
// child is pushed to the front), o0 = index of its first child with more than one key in vSizeAndPointerToNode.
__device__ __forceinline__ void emit_children(const OctLds &S, int id, int g0, int o0, int K, int freeTop, uint16_t *newList,
                                              int *sNSplit) {
    OctNode &p = S.nodes[id];
    const int hx = (p.x1 - p.x0 + 1) >> 1, hy = (p.y1 - p.y0 + 1) >> 1;   // ceil(float(d)/2)
    const int xs[3] = {p.x0, p.x0 + hx, p.x1}, ys[3] = {p.y0, p.y0 + hy, p.y1};
    uint16_t cnt[4] = {p.cnt[0], p.cnt[1], p.cnt[2], p.cnt[3]};
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int c = cnt[q];
        uint16_t cid = kNil;
        if (c != 0) {
            cid = S.freeIds[freeTop - 1 - g0];
            OctNode &ch = S.nodes[cid];
            ch.x0 = (uint16_t)xs[q & 1]; ch.x1 = (uint16_t)xs[(q & 1) + 1];
            ch.y0 = (uint16_t)ys[q >> 1]; ch.y1 = (uint16_t)ys[(q >> 1) + 1];
            ch.n = (uint16_t)c; ch.noMore = c == 1; ch.split = 0;
            ch.cnt[0] = ch.cnt[1] = ch.cnt[2] = ch.cnt[3] = 0;
            ch.child[0] = ch.child[1] = ch.child[2] = ch.child[3] = kNil;
            newList[K - 1 - g0] = cid;
            if (c > 1) S.open[o0++] = OctEntry{((uint32_t)c << 16) | ch.x0, cid, 0};
            g0++;
        }
        p.child[q] = cid;
    }
    p.split = 1;
    S.splitIds[atomicAdd(sNSplit, 1)] = (uint16_t)id;
}

__device__ __forceinline__ int quadrants_nonempty(const OctNode &nd) { return (nd.cnt[0] != 0) + (nd.cnt[1] != 0) + (nd.cnt[2] != 0) + (nd.cnt[3] != 0); }
__device__ __forceinline__ int quadrants_open(const OctNode &nd) { return (nd.cnt[0] > 1) + (nd.cnt[1] > 1) + (nd.cnt[2] > 1) + (nd.cnt[3] > 1); }

// The std::list of the reference is kept as an ARRAY in list order (front = element 0).  A pass over the list that divides
// nodes and pushes their children to the front then becomes: children (in reverse creation order) ++ surviving old nodes (in
// their old order) — two prefix sums and a scatter, done by the whole workgroup.  Only the std::sort replay of the fine
// rounds stays on one lane (its tie order is libstdc++'s introsort, orb_octree.h).
// One LDS atomic per DISTINCT key of the wave instead of one per lane: the key-points of a level arrive in cell order, so the 64 of a wave
// fall into a handful of nodes / quadrants, and in the first rounds ALL of them hit the same two or three counters (64 serialised
// updates per instruction otherwise).
// key < 0: the lane has nothing to add; lanes with equal keys must pass the same word and the same increment.
__device__ __forceinline__ void wave_agg_add(int key, unsigned int *word, unsigned int one) {
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(key >= 0);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int k0 = __builtin_amdgcn_readlane(key, leader);
        const unsigned long long mk = __ballot(key == k0);
        if (lane == leader) atomicAdd(word, one * (unsigned int)__popcll(mk));
        todo &= ~mk;
    }
}

constexpr int kKeysPerLane = 24;   // levels with at most kKeysPerLane * kOctThreads (12288) keys keep keys and owners in registers

// The keys of one level, walked by the whole workgroup with a UNIFORM trip count (wave_agg_add ballots inside the visitor).
// REG: every lane holds keys i = k * NT + tid and their owners in registers — the level is read from HBM/L2 once, with all loads
// in flight together, and the passes of the rounds below cost LDS + VALU only (a pass over keys in memory paid one dependent L2
// round trip per 512 keys: 6-10 us per pass at level 0 of a 640 x 480 frame, seven passes).  !REG: keys stay in memory, owners in owner[].
template <bool REG>
struct OctKeys {
    uint32_t ck[REG ? kKeysPerLane : 1];
    uint32_t ow[REG ? kKeysPerLane : 1];
    const uint32_t *c;
    uint16_t *own;
    int n;
    __device__ __forceinline__ void load() {
        if constexpr (REG) {
#pragma unroll
            for (int k = 0; k < kKeysPerLane; k++) {
                const int i = k * kOctThreads + (int)threadIdx.x;
                ck[k] = i < n ? c[i] : 0u;
                ow[k] = 0;
            }
        }
    }
    // f(i, valid, key, owner&): owner may be rewritten
    template <class F>
    __device__ __forceinline__ void for_each(F f) {
        if constexpr (REG) {
#pragma unroll
            for (int k = 0; k < kKeysPerLane; k++) {
                if (k * kOctThreads >= n) break;
                const int i = k * kOctThreads + (int)threadIdx.x;
                int o = (int)ow[k];
                f(i, i < n, ck[k], o);
                ow[k] = (uint32_t)o;
            }
        } else {
            for (int i0 = 0; i0 < n; i0 += 4 * kOctThreads) {            // four trips' loads in flight together
                uint32_t k4[4];
                int o4[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i = i0 + u * kOctThreads + (int)threadIdx.x;
                    k4[u] = i < n ? c[i] : 0u;
                    o4[u] = i < n ? (int)own[i] : 0;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (i0 + u * kOctThreads >= n) break;
                    const int i = i0 + u * kOctThreads + (int)threadIdx.x;
                    int o = o4[u];
                    f(i, i < n, k4[u], o);
                    if (i < n && o != o4[u]) own[i] = (uint16_t)o;
                }
            }
        }
    }
};

template <bool REG>
__device__ __forceinline__ void octree_level(const DevParams *__restrict__ P, const uint32_t *__restrict__ cand,
                                             const int32_t *__restrict__ levelStart, uint16_t *__restrict__ owner,
                                             uint32_t *__restrict__ selLevel, int32_t *__restrict__ selLevelCnt,
                                             int selLevelCap, int32_t *__restrict__ errFlag, unsigned char *lds) {
    __shared__ int sPhase, sM, sNOpen, sNFree, sNSplit, sOverflow;
    __shared__ unsigned int sRootN[kMaxRoots];
    __shared__ unsigned long long sWave[kOctThreads / 64];
    __shared__ int sSortCount[2];
    constexpr int NT = kOctThreads;

    const int tid = threadIdx.x, level = blockIdx.x, frame = blockIdx.y;
    const DevLevel &L = P->lv[level];
    const int32_t *ls = levelStart + (long long)frame * (kMaxLevels + 1);
    OctKeys<REG> keys;
    keys.n = ls[level + 1] - ls[level];
    keys.c = cand + (long long)frame * P->totalCand + ls[level];
    keys.own = owner + (long long)frame * P->totalCand + ls[level];
    const int n = keys.n;
    const uint32_t *c = keys.c;
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
    keys.load();
    const float hX = (float)W / nIni;
    const int cap = octree_pool_cap(N, nIni);

    OctLds S;
    S.nodes = reinterpret_cast<OctNode *>(lds);
    S.open = reinterpret_cast<OctEntry *>(S.nodes + cap);
    S.prev = S.open + cap;
    S.freeIds = reinterpret_cast<uint16_t *>(S.prev + cap);
    S.splitIds = S.freeIds + cap;
    S.listA = S.splitIds + cap;
    S.listB = S.listA + cap;
    S.segA = reinterpret_cast<SortSeg *>(lds + (((size_t)cap * (sizeof(OctNode) + 2 * sizeof(OctEntry) + 4 * sizeof(uint16_t)) + 7) & ~(size_t)7));
    S.segB = S.segA + (cap / 16 + 2);
    OctNode *nodes = S.nodes;
    uint16_t *A = S.listA, *B = S.listB;

    // free stack: ids cap-1 .. nIni (top of the stack = smallest id); roots take ids 0 .. nIni-1
    for (int i = tid; i < cap - nIni; i += NT) S.freeIds[i] = (uint16_t)(cap - 1 - i);
    if (tid < kMaxRoots) sRootN[tid] = 0;
    if (tid < nIni) {                              // :548-561  roots in push_back order
        OctNode &r = nodes[tid];
        r.x0 = (uint16_t)(int)(hX * (float)tid); r.x1 = (uint16_t)(int)(hX * (float)(tid + 1));
        r.y0 = 0; r.y1 = (uint16_t)Hh;
        r.next = kNil; r.prev = kNil;
        r.n = 0; r.noMore = 0; r.split = 0;
        r.cnt[0] = r.cnt[1] = r.cnt[2] = r.cnt[3] = 0;
        r.child[0] = r.child[1] = r.child[2] = r.child[3] = kNil;
    }
    if (tid == 0) { sNFree = cap - nIni; sNSplit = 0; sOverflow = 0; sPhase = 0; sNOpen = 0; }
    __syncthreads();
    // :564-567  keys -> roots, and in the same pass the quadrant populations of the roots (a root that turns out to hold one key
    // is never divided, its counts are not read)
    keys.for_each([&](int, bool valid, uint32_t ck, int &o) {
        int r = -1, q = 0;
        if (valid) {
            const int x = cand_x(ck);
            r = (int)((float)x / hX);
            o = r;
            q = oct_quadrant(nodes[r], x, cand_y(ck));
        }
        const int rr = r < 0 ? 0 : r;
        wave_agg_add(r, &sRootN[rr], 1u);
        wave_agg_add(r < 0 ? -1 : r * 4 + q, reinterpret_cast<unsigned int *>(&nodes[rr].cnt[q & 2]), 1u << (16 * (q & 1)));
    });
    __syncthreads();
    if (tid == 0) {                                // :570-578: empty roots leave the list (their ids are simply not reused)
        int m = 0;
        for (int i = 0; i < nIni; i++) {
            OctNode &r = nodes[i];
            r.n = (uint16_t)sRootN[i];
            if (r.n == 1) r.noMore = 1;
            if (r.n != 0) A[m++] = (uint16_t)i;
        }
        sM = m;
    }
    __syncthreads();

    // while (!bFinish)  :587-702
    while (true) {
        const int m = sM, phase = sPhase, nFree = sNFree;
        if (phase == 0) {
            // ---- coarse pass: every node that can be divided is divided (:590-640)
            const int chunk = (m + NT - 1) / NT, p0 = min(m, tid * chunk), p1 = min(m, p0 + chunk);
            unsigned long long loc = 0;                                     // children | kept << 20 | open children << 40
            for (int p = p0; p < p1; p++) {
                const OctNode &nd = nodes[A[p]];
                if (!nd.noMore) loc += (unsigned long long)quadrants_nonempty(nd) | ((unsigned long long)quadrants_open(nd) << 40);
                else loc += 1ull << 20;
            }
            unsigned long long tot;
            const unsigned long long base = block_scan64(loc, sWave, &tot);
            const int K = (int)(tot & 0xFFFFF), kept = (int)((tot >> 20) & 0xFFFFF), E = (int)(tot >> 40);
            if (K > nFree) {                                                 // cannot happen with cap >= 2N+16 (reported, not hidden)
                if (tid == 0) { sOverflow = 1; sPhase = 2; }
            } else {
                int g = (int)(base & 0xFFFFF), kb = (int)((base >> 20) & 0xFFFFF), o = (int)(base >> 40);
                for (int p = p0; p < p1; p++) {
                    const int id = A[p];
                    const OctNode &nd = nodes[id];
                    if (!nd.noMore) {
                        const int kc = quadrants_nonempty(nd), ko = quadrants_open(nd);
                        emit_children(S, id, g, o, K, nFree, B, &sNSplit);
                        g += kc; o += ko;
                    } else {
                        B[K + kb++] = (uint16_t)id;
                    }
                }
                if (tid == 0) {
                    const int size = K + kept;
                    sM = size; sNFree = nFree - K; sNOpen = E;
                    if (size >= N || size == m) sPhase = 2;
                    else if (size + E * 3 > N) sPhase = 1;
                }
            }
        } else {
            // ---- fine round: largest nodes first, stop as soon as there are N nodes (:646-699)
            const int nPrev = sNOpen;
            for (int i = tid; i < nPrev; i += NT) S.prev[i] = S.open[i];
            __syncthreads();
            wg_sort_like_libstdcxx(S.prev, nPrev, S.open, B, S.splitIds, S.segA, S.segB, sSortCount);   // open/B/splitIds are idle here
            // processing order t = 0.. is the sorted array walked from the back
            const int chunk = (nPrev + NT - 1) / NT, t0 = min(nPrev, tid * chunk), t1 = min(nPrev, t0 + chunk);
            unsigned long long loc = 0;                                     // children | open children << 40
            for (int t = t0; t < t1; t++) {
                const OctNode &nd = nodes[S.prev[nPrev - 1 - t].id];
                loc += (unsigned long long)quadrants_nonempty(nd) | ((unsigned long long)quadrants_open(nd) << 40);
            }
            unsigned long long tot;
            const unsigned long long base = block_scan64(loc, sWave, &tot);
            // node t is divided iff the list was still short of N before it: m + sum_{u<t}(children_u - 1) < N
            unsigned long long mine = 0;                                    // children | divided << 20 | open << 40, over MY divided nodes
            {
                int g = (int)(base & 0xFFFFF);
                for (int t = t0; t < t1; t++) {
                    const OctNode &nd = nodes[S.prev[nPrev - 1 - t].id];
                    const int kc = quadrants_nonempty(nd);
                    if (t == 0 || m + g - t < N) mine += (unsigned long long)kc | (1ull << 20) | ((unsigned long long)quadrants_open(nd) << 40);
                    g += kc;
                }
            }
            unsigned long long tot2;
            (void)block_scan64(mine, sWave, &tot2);
            const int K = (int)(tot2 & 0xFFFFF), J = (int)((tot2 >> 20) & 0xFFFFF), E = (int)(tot2 >> 40);
            if (K > nFree) {
                if (tid == 0) { sOverflow = 1; sPhase = 2; }
            } else {
                int g = (int)(base & 0xFFFFF), o = (int)(base >> 40);
                for (int t = t0; t < t1; t++) {
                    const int id = S.prev[nPrev - 1 - t].id;
                    const OctNode &nd = nodes[id];
                    const int kc = quadrants_nonempty(nd), ko = quadrants_open(nd);
                    if (t < J) emit_children(S, id, g, o, K, nFree, B, &sNSplit);
                    g += kc; o += ko;
                }
                __syncthreads();
                // the rest of the list keeps its order behind the new children
                const int chunkL = (m + NT - 1) / NT, p0 = min(m, tid * chunkL), p1 = min(m, p0 + chunkL);
                unsigned long long keep = 0;
                for (int p = p0; p < p1; p++) keep += nodes[A[p]].split ? 0 : 1;
                unsigned long long totK;
                int kb = (int)block_scan64(keep, sWave, &totK);
                for (int p = p0; p < p1; p++) {
                    const int id = A[p];
                    if (!nodes[id].split) B[K + kb++] = (uint16_t)id;
                }
                if (tid == 0) {
                    const int size = K + (int)totK;
                    sM = size; sNFree = nFree - K; sNOpen = E;
                    if (size >= N || size == m) sPhase = 2;
                }
            }
        }
        __syncthreads();
        { uint16_t *t = A; A = B; B = t; }
        if (sPhase == 2) break;                     // the last relabelling is folded into the selection pass below
        // relabel keys of divided nodes, count inside the new owners (one LDS atomic per distinct (node, quadrant) of the wave)
        keys.for_each([&](int, bool valid, uint32_t ck, int &o) {
            int key = -1, q = 0, id = 0;
            if (valid && nodes[o].split) {
                const int x = cand_x(ck), y = cand_y(ck);
                id = nodes[o].child[oct_quadrant(nodes[o], x, y)];
                o = id;
                const OctNode &nd = nodes[id];
                if (!nd.noMore) { q = oct_quadrant(nd, x, y); key = id * 4 + q; }
            }
            wave_agg_add(key, reinterpret_cast<unsigned int *>(&nodes[id].cnt[q & 2]), 1u << (16 * (q & 1)));
        });
        __syncthreads();
        {                                           // divided nodes return to the free stack
            const int ns = sNSplit, nf = sNFree;
            for (int i = tid; i < ns; i += NT) { const int id = S.splitIds[i]; nodes[id].split = 0; S.freeIds[nf + i] = (uint16_t)id; }
            __syncthreads();
            if (tid == 0) { sNFree = nf + ns; sNSplit = 0; }
        }
        __syncthreads();
    }
    // :705-721  best key of every node, nodes in list order; the keys of the nodes divided in the last round move to their
    // children on the way (best shares its word with the quadrant counts of the surviving nodes, hence the clearing first)
    const int m = sM;
    if (tid == 0 && sOverflow) atomicOr(errFlag, 2);
    for (int r = tid; r < m; r += NT) nodes[A[r]].best = 0;
    __syncthreads();
    keys.for_each([&](int i, bool valid, uint32_t ck, int &o) {
        if (valid) {
            if (nodes[o].split) o = nodes[o].child[oct_quadrant(nodes[o], cand_x(ck), cand_y(ck))];
            atomicMax(&nodes[o].best, ((uint32_t)cand_score(ck) << 16) | (uint32_t)(0xFFFF - i));
        }
    });
    __syncthreads();
    if (m > selLevelCap) {
        if (tid == 0) { *outCnt = 0; atomicOr(errFlag, 4); }
        return;
    }
    for (int r = tid; r < m; r += NT) out[r] = c[0xFFFF - (int)(nodes[A[r]].best & 0xFFFF)];
    if (tid == 0) *outCnt = m;
}

// REGS: levels of up to kKeysPerLane * kOctThreads keys run with their keys in registers (166 VGPRs: one workgroup per CU, the
// latency of one (frame, level) is what counts — small batches); !REGS: 69 VGPRs, three workgroups per CU hide each other's L2
// round trips (large batches).  launch_octree picks by the number of workgroups.
template <bool REGS>
__global__ __launch_bounds__(kOctThreads, REGS ? 1 : 8) void k_octree(const DevParams *__restrict__ P, const uint32_t *__restrict__ cand,
                                                        const int32_t *__restrict__ levelStart, uint16_t *__restrict__ owner,
                                                        uint32_t *__restrict__ selLevel, int32_t *__restrict__ selLevelCnt,
                                                        int selLevelCap, int32_t *__restrict__ errFlag) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    if constexpr (REGS) {
        const int32_t *ls = levelStart + (long long)blockIdx.y * (kMaxLevels + 1);
        if (ls[blockIdx.x + 1] - ls[blockIdx.x] <= kKeysPerLane * kOctThreads) {
            octree_level<true>(P, cand, levelStart, owner, selLevel, selLevelCnt, selLevelCap, errFlag, lds);
            return;
        }
    }
    octree_level<false>(P, cand, levelStart, owner, selLevel, selLevelCnt, selLevelCap, errFlag, lds);
}

// Concatenate levels, assign slots: in (level, list) order, key-points with lap0 <= x*scale <= lap1 fill the
// output from the back (stereoIndex--), the others from the front (monoIndex++).
__global__ __launch_bounds__(256) void k_assemble(const DevParams *__restrict__ P, const uint32_t *__restrict__ selLevel,
                                                  const int32_t *__restrict__ selLevelCnt, int selLevelCap, int lap0,
                                                  int lap1, uint32_t *__restrict__ selPacked, uint32_t *__restrict__ selMeta,
                                                  int32_t *__restrict__ selCount, int selCap, int32_t *__restrict__ countsBase, long long countsStride,
                                                  int32_t *__restrict__ errFlag) {
    __shared__ int lvStart[kMaxLevels + 1];
    __shared__ int part[256];
    const int tid = threadIdx.x, frame = blockIdx.x;
    int32_t *counts = reinterpret_cast<int32_t *>(reinterpret_cast<uint8_t *>(countsBase) + frame * countsStride);   // {n, monoIndex} of this frame
    const int nl = P->nlevels;
    if (tid == 0) {
        int run = 0;
        for (int l = 0; l < nl; l++) { lvStart[l] = run; run += selLevelCnt[(long long)frame * nl + l]; }
        lvStart[nl] = run;
    }
    __syncthreads();
    const int total = lvStart[nl];
    if (total > selCap) {
        if (tid == 0) { selCount[frame] = 0; counts[0] = total; counts[1] = 0; atomicOr(errFlag, 8); }
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
        counts[0] = total;
        counts[1] = total - run;      // monoIndex
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

// test hook: the workgroup sort on an arbitrary array (tests/test_extractor_gpu.py compares it with the real std::sort)
__global__ __launch_bounds__(kOctThreads) void k_sort_hook(OctEntry *data, int n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ int sCnt[2];
    OctEntry *a = reinterpret_cast<OctEntry *>(lds), *tmp = a + n;
    uint16_t *sf = reinterpret_cast<uint16_t *>(tmp + n), *sr = sf + n;
    SortSeg *segA = reinterpret_cast<SortSeg *>(lds + (((size_t)n * 20 + 7) & ~(size_t)7)), *segB = segA + (n / 16 + 2);
    for (int i = threadIdx.x; i < n; i += blockDim.x) a[i] = data[i];
    __syncthreads();
    wg_sort_like_libstdcxx(a, n, tmp, sf, sr, segA, segB, sCnt);
    for (int i = threadIdx.x; i < n; i += blockDim.x) data[i] = a[i];
}
int launch_sort_hook(uint32_t *keys, uint16_t *ids, int n) {
    if (n < 0 || n > 4096) return -1;
    if (n == 0) return 0;
    std::vector<OctEntry> h(n);
    for (int i = 0; i < n; i++) h[i] = OctEntry{keys[i], ids[i], 0};
    OctEntry *d = nullptr;
    if (hipMalloc((void **)&d, n * sizeof(OctEntry)) != hipSuccess) return -2;
    const size_t ldsBytes = (size_t)n * 20 + 8 + 2 * (size_t)(n / 16 + 2) * sizeof(SortSeg) + 64;
    (void)raise_lds_limit(reinterpret_cast<const void *>(k_sort_hook), ldsBytes);
    bool ok = hipMemcpy(d, h.data(), n * sizeof(OctEntry), hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(k_sort_hook, dim3(1), dim3(kOctThreads), ldsBytes, nullptr, d, n);
        ok = hipMemcpy(h.data(), d, n * sizeof(OctEntry), hipMemcpyDeviceToHost) == hipSuccess;
    }
    (void)hipFree(d);
    if (!ok) return -2;
    for (int i = 0; i < n; i++) { keys[i] = h[i].key; ids[i] = h[i].id; }
    return 0;
}

void launch_octree(const DevParams *dP, const DevParams &hP, const uint32_t *cand, const int32_t *levelStart,
                   uint16_t *owner, uint32_t *selLevel, int32_t *selLevelCnt, int selLevelCap, int32_t *errFlag,
                   int nframes, size_t ldsBytes, hipStream_t st) {
    // > 64 KiB of dynamic LDS needs the opt-in (process state that only grows: rumi_common.h)
    (void)raise_lds_limit(reinterpret_cast<const void *>(k_octree<true>), ldsBytes);
    (void)raise_lds_limit(reinterpret_cast<const void *>(k_octree<false>), ldsBytes);
    static const int forced = [] { const char *e = getenv("RUMI_OCT_REGS"); return e ? atoi(e) : -1; }();
    const bool regs = forced >= 0 ? forced != 0 : hP.nlevels * nframes <= 256;   // at most one workgroup per CU
    hipLaunchKernelGGL(regs ? k_octree<true> : k_octree<false>, dim3(hP.nlevels, nframes), dim3(kOctThreads), ldsBytes, st, dP, cand, levelStart, owner, selLevel,
                       selLevelCnt, selLevelCap, errFlag);
}
void launch_assemble(const DevParams *dP, const uint32_t *selLevel, const int32_t *selLevelCnt, int selLevelCap, int lap0,
                     int lap1, uint32_t *selPacked, uint32_t *selMeta, int32_t *selCount, int selCap, int32_t *counts, long long countsStride,
                     int32_t *errFlag, int nframes, hipStream_t st) {
    hipLaunchKernelGGL(k_assemble, dim3(nframes), dim3(256), 0, st, dP, selLevel, selLevelCnt, selLevelCap, lap0, lap1,
                       selPacked, selMeta, selCount, selCap, counts, countsStride, errFlag);
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
