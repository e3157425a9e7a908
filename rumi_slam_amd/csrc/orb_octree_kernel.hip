// On-device quadtree key-point distribution + output slot assignment (gfx950).
//
//   k_octree    one workgroup per (frame, level): DistributeOctTree, R/lib_src/ORBextractor.cc:538-724.
//               Keys stay where the FAST kernel left them (cand[], HBM/L2; in registers for a handful of frames).  The node pool, the list
//               (an array in list order), the open/sort arrays live in LDS.  The quadrant counts of the first 4-6 subdivision levels come
//               from ONE histogram pass over the keys (count tables); only a tree that outgrows them relabels keys and counts round by
//               round (a 16-bit owner id per key then, owner[]).  The list passes (divide, push children to the front, erase parents) are
//               prefix sums + scatters -- on one wave without barriers while the list is short -- and std::sort is replayed exactly
//               (wave_sort_like_libstdcxx up to 64 entries, wg_sort_like_libstdcxx beyond), so the result ORDER equals the reference's.
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
// Count tables (round 3): the key populations of EVERY cell of the first D subdivision levels, T_d[P] with P = path of the cell (root index,
// then two bits per level: P_{d+1} = 4 P_d + quadrant), d = 0 .. D.  One pass over the keys fills T_D, a sum pyramid gives the rest; after
// that a node of depth d < D is born with its quadrant counts (T_{d+1}[4 P .. 4 P + 3]) and the rounds need no pass over the keys at all.
// Level d starts at a multiple of four entries, so the four counts of a cell are one aligned 8-byte LDS read.
// D grows with the number of nodes wanted (the coarse passes stop near 4^depth = N), within 4096 cells of depth D.
__host__ __device__ inline int oct_tab_depth(int nIni, int N) {
    int D = N <= 256 ? 4 : N <= 1024 ? 5 : N <= 1152 ? 6 : 5;   // (beyond that the node pool needs the LDS)
    while (D > 1 && (nIni << (2 * D)) > 4096) D--;
    return D;
}
__host__ __device__ inline int oct_tab_off(int d, int nIni) { return d == 0 ? 0 : 16 + nIni * (((1 << (2 * d)) - 4) / 3); }
__host__ __device__ inline int oct_tab_entries(int nIni, int D) { return oct_tab_off(D + 1, nIni); }
__host__ __device__ inline int oct_leaf_entries(int nIni, int D) { return nIni << (2 * D); }
// LDS bytes of one k_octree workgroup for a pool of `cap` nodes: nodes, open + sort arrays, free stack, split stack, two list arrays,
// two SortSeg work lists, the count tables and the cell -> leaf table
// (W, H: the level's key-point area; the two per-coordinate path tables have W + 1 and H + 1 entries)
__host__ __device__ inline size_t octree_lds_bytes(int cap, int nIni, int D, int W, int H) {
    return (size_t)cap * (sizeof(OctNode) + 2 * sizeof(OctEntry) + 4 * sizeof(uint16_t)) + 2 * (size_t)(cap / 16 + 2) * 8 + 64 +
           ((size_t)(oct_tab_entries(nIni, D) + oct_leaf_entries(nIni, D) + W + H + 4) * sizeof(uint16_t) + 16);
}

// A value every lane holds (read from LDS, or the total of a scan) moves to a scalar register: the control state of the rounds would
// otherwise sit in VGPRs for the whole kernel.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned long long uni64(unsigned long long v) {
    return ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned int)__builtin_amdgcn_readfirstlane((int)v);
}

// exclusive scan of one 64-bit value per thread over the workgroup (three packed 20-bit counters); *total = sum over all threads
template <int NT>
__device__ __forceinline__ unsigned long long block_scan64(unsigned long long v, unsigned long long *sWave, unsigned long long *total) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
    for (int w = 0; w < NT / 64; w++) {
        const unsigned long long t = sWave[w];
        if (w < wave) base += t;
        tot += t;
    }
    __syncthreads();
    *total = uni64(tot);
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
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nWaves = blockDim.x >> 6;
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
        const int nSeg = uni(sCount[cur]);
        if (nSeg == 0) break;
        SortSeg *in = cur ? segB : segA, *outS = cur ? segA : segB;
        for (int si = wave; si < nSeg; si += nWaves) {
            const SortSeg sg = in[si];
            const int first = uni(sg.first), last = uni(sg.last);
            if (sg.depth == 0) {                                        // std::__partial_sort(first, last, last)
                if (lane == 0) heap_sort(a + first, a + last);
                continue;
            }
            if (lane == 0) move_median_to_first(a + first, a + first + 1, a + first + (last - first) / 2, a + last - 1);
            wave_fence_lds();
            const uint32_t piv = (uint32_t)uni((int)a[first].key);
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

// ---- the same replay for up to 64 entries, by ONE wave with the entries in registers (lane i = element i) ------------------------
// The fine rounds of a 1000-feature frame sort ~60 entries: the workgroup version spends its time in barriers and dependent LDS round
// trips (median, pivot, stop lists, swaps, segment lists: ~15 k cycles); here the stops of the two pointers are two ballots, the k-th stop
// of one pointer meets the k-th stop of the other through two ds_permute rank tables, the swap is one ds_bpermute pair, and nothing is stored in LDS.
// key / id: the lane's entry (lanes >= n: anything); sorted entries are written to out[0..n) (LDS); heapScratch: n entries of LDS
__device__ __forceinline__ void wave_sort_like_libstdcxx(uint32_t key, uint32_t id, int n, OctEntry *out, OctEntry *heapScratch) {
    using namespace sortimpl;
    const int lane = threadIdx.x & 63;
    if (n > 16) {
        int stack = 0, sp = 0;                       // a VGPR as a 64-entry array of first | last << 8 | depth << 16 (uniform)
        {
            int lg = 0;
            for (int t = n; t > 1; t >>= 1) lg++;
            if (lane == 0) stack = n << 8 | (lg * 2) << 16;
            sp = 1;
        }
        while (sp > 0) {
            const int top = __builtin_amdgcn_readlane(stack, --sp);
            int first = top & 0xFF, last = (top >> 8) & 0xFF, depth = top >> 16;
            while (last - first > 16) {
                if (depth == 0) {                    // std::__partial_sort(first, last, last): serial, through LDS (adversarial inputs only)
                    if (lane < n) heapScratch[lane] = OctEntry{key, (uint16_t)id, 0};
                    wave_fence_lds();
                    if (lane == 0) heap_sort(heapScratch + first, heapScratch + last);
                    wave_fence_lds();
                    if (lane < n) { const OctEntry e = heapScratch[lane]; key = e.key; id = e.id; }
                    wave_fence_lds();
                    break;
                }
                depth--;
                // std::__move_median_to_first(first, first + 1, mid, last - 1)
                const int ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
                const uint32_t ka = (uint32_t)__builtin_amdgcn_readlane((int)key, ia), kb = (uint32_t)__builtin_amdgcn_readlane((int)key, ib),
                               kc = (uint32_t)__builtin_amdgcn_readlane((int)key, ic);
                const int sIdx = ka < kb ? (kb < kc ? ib : (ka < kc ? ic : ia)) : (ka < kc ? ia : (kb < kc ? ic : ib));
                {
                    const uint32_t kf = (uint32_t)__builtin_amdgcn_readlane((int)key, first), idf = (uint32_t)__builtin_amdgcn_readlane((int)id, first);
                    const uint32_t ks = (uint32_t)__builtin_amdgcn_readlane((int)key, sIdx), ids = (uint32_t)__builtin_amdgcn_readlane((int)id, sIdx);
                    if (lane == first) { key = ks; id = ids; }
                    else if (lane == sIdx) { key = kf; id = idf; }
                }
                const uint32_t piv = (uint32_t)__builtin_amdgcn_readlane((int)key, first);
                // std::__unguarded_partition(first + 1, last, first): stops of the left pointer (ascending) and of the right one (descending)
                const bool inSeg = lane > first && lane < last;
                const unsigned long long MF = __ballot(inSeg && !(key < piv)), MR = __ballot(inSeg && !(piv < key));
                const int nF = __popcll(MF), nR = __popcll(MR);
                const unsigned long long below = (1ull << lane) - 1ull, above = lane == 63 ? 0ull : ~0ull << (lane + 1);
                const bool isF = (MF >> lane) & 1, isR = (MR >> lane) & 1;
                const int fBelow = __popcll(MF & below), rBelow = __popcll(MR & below);
                const int kF = fBelow, kR = __popcll(MR & above);
                // tabF[k] / tabR[k] (in lane k) = position of the k-th stop of the left / right pointer: every lane sends its index to a slot
                // of its own (stops first, by rank; the other lanes behind them), one ds_permute each
                const int tabF = __builtin_amdgcn_ds_permute((isF ? kF : nF + lane - fBelow) << 2, lane);
                const int tabR = __builtin_amdgcn_ds_permute((isR ? kR : nR + lane - rBelow) << 2, lane);
                const int rpos = __builtin_amdgcn_ds_bpermute(kF << 2, tabR), fpos = __builtin_amdgcn_ds_bpermute(kR << 2, tabF);
                const bool swF = isF && kF < nR && lane < rpos, swR = isR && kR < nF && fpos < lane;
                const int swaps = __popcll(__ballot(swF));
                {
                    const int partner = swF ? rpos : fpos;
                    const uint32_t pk = (uint32_t)__builtin_amdgcn_ds_bpermute(partner << 2, (int)key), pid = (uint32_t)__builtin_amdgcn_ds_bpermute(partner << 2, (int)id);
                    if (swF || swR) { key = pk; id = pid; }
                }
                int cut;
                if (swaps >= 1) {
                    const int nextF = swaps < nF ? __builtin_amdgcn_readlane(tabF, swaps) : 0x7FFFFFFF;
                    cut = min(nextF, __builtin_amdgcn_readlane(tabR, swaps - 1));
                } else {
                    cut = __builtin_amdgcn_readlane(tabF, 0);
                }
                if (last - cut > 16) {
                    if (lane == sp) stack = cut | last << 8 | depth << 16;
                    sp++;
                }
                last = cut;
            }
        }
    }
    // __final_insertion_sort == stable sort inside each leaf of <= 16 == stable rank over a +-15 window (elements of other leaves never
    // count: left ones are <=, right ones >=).  The neighbours come by whole-wave DPP shifts, one lane further per step.
    int pos = lane;
    {
        const uint32_t kk = lane < n ? key : 0xFFFFFFFFu;
        int l = (int)kk, r = (int)kk;
#pragma unroll
        for (int d = 1; d <= 15; d++) {
            l = __builtin_amdgcn_update_dpp(0, l, 0x138, 0xF, 0xF, false);             // wave_shr:1 -> key of lane - d (0 beyond lane 0)
            r = __builtin_amdgcn_update_dpp(-1, r, 0x130, 0xF, 0xF, false);            // wave_shl:1 -> key of lane + d (max beyond lane 63)
            pos -= (uint32_t)l > kk ? 1 : 0;
            pos += (uint32_t)r < kk ? 1 : 0;
        }
    }
    if (lane < n) out[pos] = OctEntry{key, (uint16_t)id, 0};
}

struct OctLds {
    OctNode *nodes;
    OctEntry *open, *prev;
    uint16_t *freeIds, *splitIds, *listA, *listB;
    SortSeg *segA, *segB;
    uint16_t *tab, *leaf;      // count tables T_0 .. T_D; cell of depth D -> id of the list node that covers it
    uint16_t *xbin, *ybin;     // the x half (with the root) and the y half of the depth-D path of a key, by coordinate
    int D, nIni;
};
// The kernel keeps the list as arrays, so the link fields of a node are free: they hold its place in the subdivision.
__device__ __forceinline__ uint16_t &node_path(OctNode &nd) { return nd.next; }
__device__ __forceinline__ uint16_t &node_depth(OctNode &nd) { return nd.prev; }

// DivideNode + the push_front block after it (ORBextractor.cc:471-522, :603-637) for ONE parent, given where its children land:
// g0 = creation index of its first child in this round (the list receives children in REVERSE creation order, because every
// child is pushed to the front), o0 = index of its first child with more than one key in vSizeAndPointerToNode.
// table: children of depth < D take their quadrant counts from the count tables; the return value says that a child of depth D with more
// than one key was created (its counts need a pass over the keys).  SOLO: the caller returns the parent to the free stack itself.
// Everything that is read is read first, in a handful of wide LDS loads that are in flight together: the stores may alias any of it as far
// as the compiler knows, and one dependent LDS round trip (~130 cycles) per quadrant is what the list passes used to spend their time on.
template <bool SOLO>
__device__ __forceinline__ bool emit_children(const OctLds &S, int id, int g0, int o0, int K, int freeTop, uint16_t *newList,
                                              int *sNSplit, bool table) {
    OctNode &p = S.nodes[id];
    const uint4 plo = reinterpret_cast<const uint4 *>(&p)[0];               // x0 y0 | x1 y1 | path depth | n noMore split
    const uint2 pcn = *reinterpret_cast<const uint2 *>(&p.cnt[0]);
    const int x0 = (int)(plo.x & 0xFFFFu), y0 = (int)(plo.x >> 16), x1 = (int)(plo.y & 0xFFFFu), y1 = (int)(plo.y >> 16);
    const int dc = (int)(plo.z >> 16) + 1, pc0 = (int)(plo.z & 0xFFFFu) * 4;
    const int hx = (x1 - x0 + 1) >> 1, hy = (y1 - y0 + 1) >> 1;            // ceil(float(d)/2)
    const int xs[3] = {x0, x0 + hx, x1}, ys[3] = {y0, y0 + hy, y1};
    const int cnt[4] = {(int)(pcn.x & 0xFFFFu), (int)(pcn.x >> 16), (int)(pcn.y & 0xFFFFu), (int)(pcn.y >> 16)};
    const bool born = table && dc < S.D;                                   // children come with their counts
    uint16_t fid[4];
    uint2 cc[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        fid[j] = S.freeIds[max(freeTop - 1 - g0 - j, 0)];
        cc[j] = born ? *reinterpret_cast<const uint2 *>(S.tab + oct_tab_off(dc + 1, S.nIni) + 4 * (pc0 + j)) : make_uint2(0u, 0u);
    }
    bool deep = false;
    int made = 0;
    uint32_t kids[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int c = cnt[q];
        if (c != 0) {
            const uint16_t cid = made == 0 ? fid[0] : made == 1 ? fid[1] : made == 2 ? fid[2] : fid[3];
            const uint32_t cx0 = (uint32_t)xs[q & 1], cx1 = (uint32_t)xs[(q & 1) + 1], cy0 = (uint32_t)ys[q >> 1], cy1 = (uint32_t)ys[(q >> 1) + 1];
            uint4 *ch = reinterpret_cast<uint4 *>(&S.nodes[cid]);
            ch[0] = make_uint4(cx0 | (cy0 << 16), cx1 | (cy1 << 16), (uint32_t)(pc0 + q) | ((uint32_t)dc << 16), (uint32_t)c | (c == 1 ? 0x10000u : 0u));
            ch[1] = make_uint4(c > 1 ? cc[q].x : 0u, c > 1 ? cc[q].y : 0u, 0xFFFFFFFFu, 0xFFFFFFFFu);
            deep |= table && c > 1 && dc >= S.D;
            newList[K - 1 - g0 - made] = cid;
            if (c > 1) S.open[o0++] = OctEntry{((uint32_t)c << 16) | cx0, cid, 0};
            kids[q >> 1] = (kids[q >> 1] & ~(0xFFFFu << (16 * (q & 1)))) | ((uint32_t)cid << (16 * (q & 1)));
            made++;
        }
    }
    *reinterpret_cast<uint2 *>(&p.child[0]) = make_uint2(kids[0], kids[1]);
    if constexpr (!SOLO) {
        p.split = 1;
        S.splitIds[atomicAdd(sNSplit, 1)] = (uint16_t)id;
    }
    return deep;
}

__device__ __forceinline__ int quadrants_nonempty(const OctNode &nd) { return (nd.cnt[0] != 0) + (nd.cnt[1] != 0) + (nd.cnt[2] != 0) + (nd.cnt[3] != 0); }
__device__ __forceinline__ int quadrants_open(const OctNode &nd) { return (nd.cnt[0] > 1) + (nd.cnt[1] > 1) + (nd.cnt[2] > 1) + (nd.cnt[3] > 1); }

// The std::list of the reference is kept as an ARRAY in list order (front = element 0).  A pass over the list that divides
// nodes and pushes their children to the front then becomes: children (in reverse creation order) ++ surviving old nodes (in
// their old order) — two prefix sums and a scatter.
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

// The same for keys that arrive in RUNS (the cells of a level in raster order, the keys of a cell row by row: neighbouring lanes fall into the
// same cell of the subdivision): the first lane of every run of equal keys adds the length of its run.  No loop, one LDS atomic per run.
__device__ __forceinline__ void wave_run_add(int key, unsigned int *word, unsigned int one) {
    const int lane = threadIdx.x & 63;
    const int before = __shfl_up(key, 1);
    const bool head = lane == 0 || key != before;
    const unsigned long long heads = __ballot(head);
    const unsigned long long rest = lane == 63 ? 0ull : heads >> (lane + 1);
    const int len = rest ? __ffsll((long long)rest) : 64 - lane;
    if (head && key >= 0) atomicAdd(word, one * (unsigned int)len);
}

// inclusive prefix sum over the wave in six DPP steps (no LDS round trips: __shfl_up is a ds_bpermute each)
__device__ __forceinline__ unsigned int wave_scan_incl(unsigned int v) {
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);      // row_shr:1
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);      // row_shr:2
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);      // row_shr:4
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);      // row_shr:8
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);      // row_bcast:15 -> rows 1, 3
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);      // row_bcast:31 -> rows 2, 3
    return v;
}

// atomicMax(best[owner], v) for the lanes of a wave, one atomic per run of equal owners inside a row of 16 lanes (row-wide DPP shifts: the
// last lane of a run ends up with the run's maximum and is the only one that goes to LDS; 64 lanes on two or three counters serialise otherwise).
#define RUMI_RUN_MAX_STEP(S_)                                                              \
    {                                                                                      \
        const int o2 = __builtin_amdgcn_update_dpp(-2, oo, 0x110 + S_, 0xF, 0xF, false);   \
        const int v2 = __builtin_amdgcn_update_dpp(0, (int)v, 0x110 + S_, 0xF, 0xF, false); \
        if (o2 == oo) v = max(v, (uint32_t)v2);                                            \
    }
__device__ __forceinline__ void wave_run_max(OctNode *nodes, int owner, bool valid, uint32_t v) {
    const int oo = valid ? owner : -1;
    RUMI_RUN_MAX_STEP(1) RUMI_RUN_MAX_STEP(2) RUMI_RUN_MAX_STEP(4) RUMI_RUN_MAX_STEP(8)
    const int after = __builtin_amdgcn_update_dpp(-2, oo, 0x101, 0xF, 0xF, false);          // row_shl:1 = the next lane's owner
    if (valid && after != oo) atomicMax(&nodes[owner].best, v);
}
#undef RUMI_RUN_MAX_STEP

constexpr int kKeysPerLane = 24;   // levels with at most kKeysPerLane * kOctThreads (12288) keys keep keys and owners in registers

// The keys of one level, walked by the whole workgroup with a UNIFORM trip count (wave_agg_add ballots inside the visitor).
// REG: every lane holds keys i = k * NT + tid and their owners in registers — the level is read from HBM/L2 once, with all loads
// in flight together, and the passes of the rounds below cost LDS + VALU only (a pass over keys in memory paid one dependent L2
// round trip per 512 keys: 6-10 us per pass at level 0 of a 640 x 480 frame, seven passes).  !REG: keys stay in memory, owners in owner[].
template <bool REG, int NT>
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
                const int i = k * NT + (int)threadIdx.x;
                ck[k] = i < n ? c[i] : 0u;
                ow[k] = 0;
            }
        }
    }
    // f(i, valid, key, owner&): owner may be rewritten.  OWN (keys in memory only): 0 = owners are read and written back when they change,
    // 1 = no owner traffic at all (the visitor gets 0 and what it leaves is dropped), 2 = written, not read.
    template <int OWN = 0, class F>
    __device__ __forceinline__ void for_each(F f) {
        if constexpr (REG) {
#pragma unroll
            for (int k = 0; k < kKeysPerLane; k++) {
                if (k * NT >= n) break;
                const int i = k * NT + (int)threadIdx.x;
                int o = (int)ow[k];
                f(i, i < n, ck[k], o);
                ow[k] = (uint32_t)o;
            }
        } else {
            for (int i0 = 0; i0 < n; i0 += 4 * NT) {            // four trips' loads in flight together
                uint32_t k4[4];
                int o4[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i = i0 + u * NT + (int)threadIdx.x;
                    k4[u] = i < n ? c[i] : 0u;
                    o4[u] = OWN == 0 && i < n ? (int)own[i] : 0;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (i0 + u * NT >= n) break;
                    const int i = i0 + u * NT + (int)threadIdx.x;
                    int o = o4[u];
                    f(i, i < n, k4[u], o);
                    if (OWN != 1 && i < n && (OWN == 2 || o != o4[u])) own[i] = (uint16_t)o;
                }
            }
        }
    }
};

template <bool REG, int NT>
__device__ __forceinline__ void octree_level(const DevParams *__restrict__ P, const uint32_t *__restrict__ cand,
                                             const int32_t *__restrict__ levelStart, uint16_t *__restrict__ owner,
                                             uint32_t *__restrict__ selLevel, int32_t *__restrict__ selLevelCnt,
                                             int selLevelCap, int32_t *__restrict__ errFlag, unsigned char *lds) {
    __shared__ int sPhase, sM, sNOpen, sNFree, sNSplit, sOverflow, sDeep, sSolo;
    __shared__ unsigned long long sWave[NT / 64];
    __shared__ int sSortCount[2];

    const int tid = threadIdx.x, level = blockIdx.x, frame = blockIdx.y;
    const DevLevel &L = P->lv[level];
    const int32_t *ls = levelStart + (long long)frame * (kMaxLevels + 1);
    OctKeys<REG, NT> keys;
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
    const int nIni = uni((int)__builtin_roundf((float)W / (float)Hh));   // (float arithmetic is per-lane: everything derived from it would be too)
    if (nIni <= 0 || nIni > kMaxRoots) {           // the reference divides by zero / we do not stage that many roots
        if (tid == 0) { *outCnt = 0; atomicOr(errFlag, 1); }
        return;
    }
#ifdef RUMI_OCT_STAMP
    long long stAcc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stLast = clock64(); int stRounds[2] = {0, 0};
#define OCT_STAMP(k) do { if (tid == 0) { const long long t_ = clock64(); stAcc[k] += t_ - stLast; stLast = t_; } } while (0)
#else
#define OCT_STAMP(k) do { } while (0)
#endif
    keys.load();
    const float hX = __int_as_float(uni(__float_as_int((float)W / nIni)));
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
    S.D = oct_tab_depth(nIni, N); S.nIni = nIni;
    S.tab = reinterpret_cast<uint16_t *>((reinterpret_cast<uintptr_t>(S.segB + (cap / 16 + 2)) + 7) & ~(uintptr_t)7);
    S.leaf = S.tab + oct_tab_entries(nIni, S.D);
    S.xbin = S.leaf + oct_leaf_entries(nIni, S.D);
    S.ybin = S.xbin + W + 1;
    OctNode *nodes = S.nodes;
    uint16_t *A = S.listA, *B = S.listB;
    const int D = S.D, offD = oct_tab_off(D, nIni);

    // free stack: ids cap-1 .. nIni (top of the stack = smallest id); roots take ids 0 .. nIni-1
    for (int i = tid; i < cap - nIni; i += NT) S.freeIds[i] = (uint16_t)(cap - 1 - i);
    for (int i = tid; i < (nIni << (2 * D)) / 2; i += NT) reinterpret_cast<uint32_t *>(S.tab + offD)[i] = 0u;
    if (tid == 0) { sNFree = cap - nIni; sNSplit = 0; sOverflow = 0; sPhase = 0; sNOpen = 0; sDeep = 0; }
    for (int i = tid; i <= W + Hh + 1; i += NT) {  // per-coordinate halves of the path: bit 2 (D-1-d) (x) / 2 (D-1-d) + 1 (y) = side taken at depth d
        const bool isX = i <= W;
        const int v = isX ? i : i - (W + 1);
        const int r = isX ? min((int)((float)v / hX), nIni - 1) : 0;   // (v = W itself is not a key position)
        int lo = isX ? (int)(hX * (float)r) : 0, hi = isX ? (int)(hX * (float)(r + 1)) : Hh;
        int bits = 0;
        for (int d = 0; d < D; d++) {
            const int mid = lo + ((hi - lo + 1) >> 1);
            const bool q = v >= mid;
            lo = q ? mid : lo; hi = q ? hi : mid;
            bits = bits * 4 + (q ? 1 : 0);
        }
        if (isX) S.xbin[v] = (uint16_t)((r << (2 * D)) | bits);
        else S.ybin[v] = (uint16_t)(bits << 1);
    }
    __syncthreads();
    OCT_STAMP(7);
    // :564-567  keys -> roots, and with the same comparisons DivideNode would make (:471-511) on down to depth D: the key's cell there.
    // The x comparisons of a path depend on x alone (the root is a function of x) and the y comparisons on y alone, so the two halves of
    // the path come from two small tables by coordinate.  A key keeps the path of its cell (its "owner" until the tree outgrows the
    // tables); T_D counts the keys of every cell.
    auto cell_of = [&](uint32_t ck) { return (int)S.xbin[min(cand_x(ck), W)] | (int)S.ybin[min(cand_y(ck), Hh)]; };
    keys.template for_each<1>([&](int, bool valid, uint32_t ck, int &o) {      // (keys in memory: the cell is looked up again when it is needed)
        int P = -1;
        if (valid) {
            P = cell_of(ck);
            o = P;
        }
        wave_run_add(P, reinterpret_cast<unsigned int *>(S.tab + offD) + ((P < 0 ? 0 : P) >> 1), 1u << (16 * (P & 1)));
    });
    __syncthreads();
    OCT_STAMP(8);
    for (int d = D - 1; d >= 0; d--) {             // sum pyramid
        const uint16_t *src = S.tab + oct_tab_off(d + 1, nIni);
        uint16_t *dst = S.tab + oct_tab_off(d, nIni);
        for (int e = tid; e < (nIni << (2 * d)); e += NT) {
            const uint2 c4 = *reinterpret_cast<const uint2 *>(src + 4 * e);
            dst[e] = (uint16_t)((c4.x & 0xFFFFu) + (c4.x >> 16) + (c4.y & 0xFFFFu) + (c4.y >> 16));
        }
        __syncthreads();
    }
    if (tid < nIni) {                              // :548-561  roots in push_back order
        OctNode &r = nodes[tid];
        r.x0 = (uint16_t)(int)(hX * (float)tid); r.x1 = (uint16_t)(int)(hX * (float)(tid + 1));
        r.y0 = 0; r.y1 = (uint16_t)Hh;
        node_depth(r) = 0; node_path(r) = (uint16_t)tid;
        r.n = S.tab[tid]; r.noMore = r.n == 1; r.split = 0;
        *reinterpret_cast<uint2 *>(&r.cnt[0]) = *reinterpret_cast<const uint2 *>(S.tab + oct_tab_off(1, nIni) + 4 * tid);
        r.child[0] = r.child[1] = r.child[2] = r.child[3] = kNil;
    }
    __syncthreads();
    if (tid == 0) {                                // :570-578: empty roots leave the list (their ids are simply not reused)
        int m = 0;
        for (int i = 0; i < nIni; i++)
            if (nodes[i].n != 0) A[m++] = (uint16_t)i;
        sM = m;
    }
    __syncthreads();
    bool table = true;                             // uniform: the count tables still cover every open node
    // leaf[cell of depth D] = the node of `list` that covers it; the cells of a node are consecutive paths.  One thread per node writes up
    // to 16 cells itself; the few shallower nodes (64, 256 cells: sparse levels) are filled by their wave, lane per cell.
    auto fill_leaf = [&](const uint16_t *list, int len) {
        for (int r0 = 0; r0 < len; r0 += NT) {
            const int r = r0 + tid;
            int id = 0, span = 0, base = 0;
            if (r < len) {
                id = list[r];
                OctNode &nd = nodes[id];
                span = 1 << (2 * (D - (int)node_depth(nd)));
                base = (int)node_path(nd) * span;
                if (span <= 16)
                    for (int cI = 0; cI < span; cI++) S.leaf[base + cI] = (uint16_t)id;
            }
            unsigned long long big = __ballot(span > 16);
            while (big) {
                const int src = __ffsll((long long)big) - 1;
                big &= big - 1;
                const int bId = __builtin_amdgcn_readlane(id, src), bSpan = __builtin_amdgcn_readlane(span, src), bBase = __builtin_amdgcn_readlane(base, src);
                for (int cI = tid & 63; cI < bSpan; cI += 64) S.leaf[bBase + cI] = (uint16_t)bId;
            }
        }
    };
    OCT_STAMP(0);
    // while (!bFinish)  :587-702
    while (true) {
        const int m = uni(sM), phase = uni(sPhase), nFree = uni(sNFree);
#ifdef RUMI_OCT_STAMP
        stRounds[phase != 0]++;
#endif
        bool freed = false;
        if (table && phase == 0 && m <= 64) {
            // ---- the first coarse passes, while the list fits one wave and children are born with their counts: wave 0 runs them back to
            // back (divide, push to the front, free the parents) on wave-level scans, without a workgroup barrier; the others wait below
            if (tid < 64) {
                int mm = m, nf = nFree, ph = 0, rounds = 0, E = 0;
                bool anyDeep = false;
                uint16_t *La = A, *Lb = B;
                while (true) {
                    const bool has = tid < mm;
                    int id = 0;
                    unsigned int loc = 0;                                   // children | kept << 10 | open children << 20
                    bool div = false;
                    if (has) {
                        id = La[tid];
                        const OctNode &nd = nodes[id];
                        div = !nd.noMore;
                        loc = div ? (unsigned int)quadrants_nonempty(nd) | ((unsigned int)quadrants_open(nd) << 20) : 1u << 10;
                    }
                    const unsigned int inc = wave_scan_incl(loc);
                    const unsigned int tot = (unsigned int)__builtin_amdgcn_readlane((int)inc, 63);
                    const int K = (int)(tot & 0x3FF), kept = (int)((tot >> 10) & 0x3FF);
                    if (K > nf) {                                           // cannot happen with cap >= 2N+16 (reported, not hidden)
                        if (tid == 0) sOverflow = 1;
                        ph = 2;
                        break;
                    }
                    E = (int)(tot >> 20);
                    const unsigned int base = inc - loc;
                    bool deep = false;
                    if (has) {
                        if (div) deep = emit_children<true>(S, id, (int)(base & 0x3FF), (int)(base >> 20), K, nf, Lb, nullptr, true);
                        else Lb[K + (int)((base >> 10) & 0x3FF)] = (uint16_t)id;
                    }
                    // divided nodes return to the free stack (the slots their children came from: every read of those is already issued)
                    const unsigned long long dv = __ballot(div);
                    nf -= K;
                    if (div) S.freeIds[nf + __popcll(dv & ((1ull << tid) - 1ull))] = (uint16_t)id;
                    nf += __popcll(dv);
                    if (__ballot(deep)) anyDeep = true;
                    wave_fence_lds();
                    const int size = K + kept;
                    ph = (size >= N || size == mm) ? 2 : (size + E * 3 > N ? 1 : 0);
                    { uint16_t *t = La; La = Lb; Lb = t; }
                    rounds++;
                    mm = size;
                    if (ph != 0 || mm > 64 || anyDeep) break;
                }
                if (tid == 0 && anyDeep) sDeep = 1;
                if (tid == 0) { sM = mm; sNFree = nf; sNOpen = E; sPhase = ph; sSolo = rounds; }
            }
            __syncthreads();
            if (uni(sSolo) & 1) { uint16_t *t = A; A = B; B = t; }
            freed = true;
        } else {
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
            const unsigned long long base = block_scan64<NT>(loc, sWave, &tot);
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
                        if (emit_children<false>(S, id, g, o, K, nFree, B, &sNSplit, table)) sDeep = 1;
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
            const int nPrev = uni(sNOpen);
            if (nPrev <= 64) {                      // one wave, entries in registers
                if (tid < 64) {
                    OctEntry e = OctEntry{0u, 0, 0};
                    if (tid < nPrev) e = S.open[tid];
                    wave_fence_lds();
                    wave_sort_like_libstdcxx(e.key, e.id, nPrev, S.prev, S.open);
                }
                __syncthreads();
            } else {
                for (int i = tid; i < nPrev; i += NT) S.prev[i] = S.open[i];
                __syncthreads();
                wg_sort_like_libstdcxx(S.prev, nPrev, S.open, B, S.splitIds, S.segA, S.segB, sSortCount);   // open/B/splitIds are idle here
            }
            OCT_STAMP(2);
            // processing order t = 0.. is the sorted array walked from the back
            const int chunk = (nPrev + NT - 1) / NT, t0 = min(nPrev, tid * chunk), t1 = min(nPrev, t0 + chunk);
            unsigned long long loc = 0;                                     // children | open children << 40
            for (int t = t0; t < t1; t++) {
                const OctNode &nd = nodes[S.prev[nPrev - 1 - t].id];
                loc += (unsigned long long)quadrants_nonempty(nd) | ((unsigned long long)quadrants_open(nd) << 40);
            }
            unsigned long long tot;
            const unsigned long long base = block_scan64<NT>(loc, sWave, &tot);
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
            (void)block_scan64<NT>(mine, sWave, &tot2);
            const int K = (int)(tot2 & 0xFFFFF), J = (int)((tot2 >> 20) & 0xFFFFF), E = (int)(tot2 >> 40);
            if (K > nFree) {
                if (tid == 0) { sOverflow = 1; sPhase = 2; }
            } else {
                int g = (int)(base & 0xFFFFF), o = (int)(base >> 40);
                for (int t = t0; t < t1; t++) {
                    const int id = S.prev[nPrev - 1 - t].id;
                    const OctNode &nd = nodes[id];
                    const int kc = quadrants_nonempty(nd), ko = quadrants_open(nd);
                    if (t < J) if (emit_children<false>(S, id, g, o, K, nFree, B, &sNSplit, table)) sDeep = 1;
                    g += kc; o += ko;
                }
                __syncthreads();
                // the rest of the list keeps its order behind the new children
                const int chunkL = (m + NT - 1) / NT, p0 = min(m, tid * chunkL), p1 = min(m, p0 + chunkL);
                unsigned long long keep = 0;
                for (int p = p0; p < p1; p++) keep += nodes[A[p]].split ? 0 : 1;
                unsigned long long totK;
                int kb = (int)block_scan64<NT>(keep, sWave, &totK);
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
        }
        OCT_STAMP(phase == 0 ? 1 : 3);
        if (uni(sPhase) == 2) break;                // the last relabelling is folded into the selection pass below
        const int m2 = uni(sM);
        if (table) {
            // Nothing to do while the children were born with their counts.  Once a cell of depth D has been opened the tables end:
            // every key moves from its cell to the node that covers the cell now, the keys of those deepest nodes are counted, and the
            // rounds after this one relabel and count as they go (below).
            if (uni(sDeep)) {
                fill_leaf(A, m2);
                __syncthreads();
                keys.template for_each<2>([&](int, bool valid, uint32_t ck, int &o) {
                    int key = -1, q = 0, id = 0;
                    if (valid) {
                        id = S.leaf[REG ? o : cell_of(ck)];
                        o = id;
                        OctNode &nd = nodes[id];
                        if (!nd.noMore && node_depth(nd) == D) { q = oct_quadrant(nd, cand_x(ck), cand_y(ck)); key = id * 4 + q; }
                    }
                    wave_agg_add(key, reinterpret_cast<unsigned int *>(&nodes[id].cnt[q & 2]), 1u << (16 * (q & 1)));
                });
                table = false;
            }
        } else {
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
        }
        __syncthreads();
        OCT_STAMP(4);
        if (!freed) {                               // divided nodes return to the free stack
            const int ns = uni(sNSplit), nf = uni(sNFree);
            for (int i = tid; i < ns; i += NT) { const int id = S.splitIds[i]; nodes[id].split = 0; S.freeIds[nf + i] = (uint16_t)id; }
            __syncthreads();
            if (tid == 0) { sNFree = nf + ns; sNSplit = 0; }
        }
        __syncthreads();
        OCT_STAMP(5);
    }
    // :705-721  best key of every node, nodes in list order; the keys of the nodes divided in the last round move to their
    // children on the way (best shares its word with the quadrant counts of the surviving nodes, hence the clearing first)
    const int m = uni(sM);
    if (tid == 0 && sOverflow) atomicOr(errFlag, 2);
    if (table) fill_leaf(A, m);                    // every node of the final list is at most D deep here
    for (int r = tid; r < m; r += NT) nodes[A[r]].best = 0;
    __syncthreads();
    OCT_STAMP(9);
    if (table) {
        keys.template for_each<1>([&](int i, bool valid, uint32_t ck, int &o) {
            wave_run_max(nodes, valid ? (int)S.leaf[REG ? o : cell_of(ck)] : 0, valid, ((uint32_t)cand_score(ck) << 16) | (uint32_t)(0xFFFF - i));
        });
    } else {
        keys.for_each([&](int i, bool valid, uint32_t ck, int &o) {
            if (valid && nodes[o].split) o = nodes[o].child[oct_quadrant(nodes[o], cand_x(ck), cand_y(ck))];
            wave_run_max(nodes, o, valid, ((uint32_t)cand_score(ck) << 16) | (uint32_t)(0xFFFF - i));
        });
    }
    __syncthreads();
    OCT_STAMP(10);
    if (m > selLevelCap) {
        if (tid == 0) { *outCnt = 0; atomicOr(errFlag, 4); }
        return;
    }
    for (int r = tid; r < m; r += NT) out[r] = c[0xFFFF - (int)(nodes[A[r]].best & 0xFFFF)];
    if (tid == 0) *outCnt = m;
    OCT_STAMP(6);
#ifdef RUMI_OCT_STAMP
    if (tid == 0 && frame == 0) printf("oct L%d n=%d N=%d m=%d D=%d %s rounds %d+%d  setup %lld coarse %lld sort %lld fine %lld relabel %lld free %lld select %lld | init %lld pass1 %lld leaf %lld pass2 %lld\n", level, n, N, m, D, table ? "tables" : "relabel", stRounds[0], stRounds[1], stAcc[0], stAcc[1], stAcc[2], stAcc[3], stAcc[4], stAcc[5], stAcc[6], stAcc[7], stAcc[8], stAcc[9], stAcc[10]);
#endif
}

// REGS: levels of up to kKeysPerLane * kOctThreads keys run with their keys in registers (one workgroup per CU, the latency of one
// (frame, level) is what counts -- small batches); !REGS: 120 VGPRs, two workgroups per CU hide each other's LDS and L2 round trips (large
// batches; at 64 VGPRs the loads-first emit_children spills and the launch is 20-60 % slower).  launch_octree picks by the number of workgroups.
// Large batches: NT = 256 up to ~1200 features per frame (four workgroups per CU overlap their serial phases), 512 beyond (the passes over
// the keys weigh more).
template <bool REGS, int NT>
__global__ __launch_bounds__(NT, REGS ? 1 : 4) void k_octree(const DevParams *__restrict__ P, const uint32_t *__restrict__ cand,
                                                        const int32_t *__restrict__ levelStart, uint16_t *__restrict__ owner,
                                                        uint32_t *__restrict__ selLevel, int32_t *__restrict__ selLevelCnt,
                                                        int selLevelCap, int32_t *__restrict__ errFlag) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    if constexpr (REGS) {
        const int32_t *ls = levelStart + (long long)blockIdx.y * (kMaxLevels + 1);
        if (ls[blockIdx.x + 1] - ls[blockIdx.x] <= kKeysPerLane * NT) {
            octree_level<true, NT>(P, cand, levelStart, owner, selLevel, selLevelCnt, selLevelCap, errFlag, lds);
            return;
        }
    }
    octree_level<false, NT>(P, cand, levelStart, owner, selLevel, selLevelCnt, selLevelCap, errFlag, lds);
}

// Concatenate levels, assign slots: in (level, list) order, key-points with lap0 <= x*scale <= lap1 fill the
// output from the back (stereoIndex--), the others from the front (monoIndex++).
__global__ __launch_bounds__(256) void k_assemble(const DevParams *__restrict__ P, const uint32_t *__restrict__ selLevel,
                                                  const int32_t *__restrict__ selLevelCnt, int selLevelCap, int lap0,
                                                  int lap1, uint32_t *__restrict__ selPacked, uint32_t *__restrict__ selMeta,
                                                  int32_t *__restrict__ selCount, int selCap, int32_t *__restrict__ countsBase, long long countsStride,
                                                  int32_t *__restrict__ errFlag, int32_t *__restrict__ errMirror) {
    // errMirror (one-frame calls whose results go straight to pinned host memory): the call's error word is final when this workgroup ends -- every
    // kernel that can set a bit has run, the descriptor kernel sets none -- and is published beside the results: no copy of it follows
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
        if (tid == 0) { selCount[frame] = 0; counts[0] = total; counts[1] = 0; const int old = atomicOr(errFlag, 8); if (errMirror) *errMirror = old | 8; }
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
        if (errMirror) *errMirror = *errFlag;
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
    if (n <= 64) {                                  // as the fine rounds of octree_level choose
        if (threadIdx.x < 64) {
            OctEntry e = OctEntry{0u, 0, 0};
            if ((int)threadIdx.x < n) e = a[threadIdx.x];
            wave_fence_lds();
            wave_sort_like_libstdcxx(e.key, e.id, n, tmp, a);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += blockDim.x) a[i] = tmp[i];
        __syncthreads();
    } else {
        wg_sort_like_libstdcxx(a, n, tmp, sf, sr, segA, segB, sCnt);
    }
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
    static const int forced = [] { const char *e = getenv("RUMI_OCT_REGS"); return e ? atoi(e) : -1; }();
    const bool regs = forced >= 0 ? forced != 0 : hP.nlevels * nframes <= 256;   // at most one workgroup per CU
    auto go = [&](auto kern, int threads) {
        // > 64 KiB of dynamic LDS needs the opt-in (process state that only grows: rumi_common.h)
        (void)raise_lds_limit(reinterpret_cast<const void *>(kern), ldsBytes);
        hipLaunchKernelGGL(kern, dim3(hP.nlevels, nframes), dim3(threads), ldsBytes, st, dP, cand, levelStart, owner, selLevel, selLevelCnt, selLevelCap,
                           errFlag);
    };
    if (regs) go(k_octree<true, kOctThreads>, kOctThreads);
    else if (hP.lv[0].nfeat <= 260) go(k_octree<false, 256>, 256);             // level 0 of 1200 features at scale 1.2, 8 levels
    else go(k_octree<false, kOctThreads>, kOctThreads);
}
void launch_assemble(const DevParams *dP, const uint32_t *selLevel, const int32_t *selLevelCnt, int selLevelCap, int lap0,
                     int lap1, uint32_t *selPacked, uint32_t *selMeta, int32_t *selCount, int selCap, int32_t *counts, long long countsStride,
                     int32_t *errFlag, int nframes, hipStream_t st, int32_t *errMirror) {
    hipLaunchKernelGGL(k_assemble, dim3(nframes), dim3(256), 0, st, dP, selLevel, selLevelCnt, selLevelCap, lap0, lap1,
                       selPacked, selMeta, selCount, selCap, counts, countsStride, errFlag, nframes == 1 ? errMirror : nullptr);
}
size_t octree_lds_for(const DevParams &hP) {
    size_t mx = 0;
    for (int l = 0; l < hP.nlevels; l++) {
        const int W = hP.lv[l].maxBX - kBorder, Hh = hP.lv[l].maxBY - kBorder;
        int nIni = (int)__builtin_roundf((float)W / (float)Hh);
        if (nIni < 1) nIni = 1;
        const int nr = nIni > kMaxRoots ? kMaxRoots : nIni;
        mx = std::max(mx, octree_lds_bytes(octree_pool_cap(hP.lv[l].nfeat, nIni), nr, oct_tab_depth(nr, hP.lv[l].nfeat), W, Hh));
    }
    return mx;
}

}  // namespace rumi
