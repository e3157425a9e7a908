// Quadtree ("OctTree") key-point distribution — array-based restatement of
// ORBextractor::DistributeOctTree / ExtractorNode::DivideNode / compareNodes
// (R/lib_src/ORBextractor.cc:471-724): node / entry types, the serial replay of libstdc++'s std::sort and a serial host
// driver of the algorithm (test hooks).  The HIP kernel (orb_octree_kernel.hip) shares the types and the helpers and runs the
// list passes and the sort workgroup-parallel.
//
// Design (MI355X-first, not the reference's std::list<ExtractorNode> with per-node key vectors):
//   * keys never move; every key carries the id of the node that currently owns it (owner[i]);
//   * a node is 32 bytes of LDS: rectangle, list links, the key counts of its four quadrants, child ids;
//   * one "sweep" over the keys relabels them after a round of splits and counts quadrant populations
//     of the new nodes (a plain loop in the host driver below; the kernel needs it only for trees deeper than its
//     count tables, which give the counts of the first subdivision levels from one histogram pass);
//   * the list choreography (push_front / erase / walk order), the (size, UL.x) sort and the early
//     break are a serial step driven only by those counts, so the result ORDER is the reference's.
// The reference's result depends on the tie order of libstdc++'s std::sort; sort_like_libstdcxx()
// below replays that algorithm (introsort, threshold 16, median-of-3 to first, unguarded partition,
// heap-sort fallback, final insertion sort) on (key, id) pairs so ties land where std::sort puts them.
#pragma once
#include <cstdint>
#include <vector>

#if defined(__HIPCC__)
#define RUMI_HD __host__ __device__ inline
#else
#define RUMI_HD inline
#endif

namespace rumi {

constexpr uint16_t kNil = 0xFFFF;

struct OctNode {
    uint16_t x0, y0, x1, y1;   // [x0,x1) x [y0,y1) — UL=(x0,y0) UR=(x1,y0) BL=(x0,y1) BR=(x1,y1)
    uint16_t next, prev;       // list links (kNil = end)
    uint16_t n;                // number of keys
    uint8_t noMore;            // bNoMore
    uint8_t split;             // set once the node has been divided (its keys await relabelling)
    union {
        uint16_t cnt[4];       // keys per quadrant (n1..n4), valid when n > 1, until the tree is final
        uint32_t best;         // final pass only: maximum of (response << 16 | ~index) over the node's keys
    };
    uint16_t child[4];         // ids of the children after a split (kNil = empty child)
};
static_assert(sizeof(OctNode) == 32, "OctNode is one 32-byte LDS record");

struct OctEntry {              // one element of vSizeAndPointerToNode
    uint32_t key;              // (size << 16) | UL.x  -> compareNodes is a plain integer compare
    uint16_t id;
    uint16_t pad;
};

// ---- libstdc++ std::sort replay --------------------------------------------------------------
namespace sortimpl {
RUMI_HD bool lt(const OctEntry &a, const OctEntry &b) { return a.key < b.key; }
RUMI_HD void swp(OctEntry &a, OctEntry &b) { OctEntry t = a; a = b; b = t; }

RUMI_HD void unguarded_linear_insert(OctEntry *last) {
    OctEntry val = *last;
    OctEntry *next = last - 1;
    while (lt(val, *next)) { *last = *next; last = next; --next; }
    *last = val;
}
RUMI_HD void insertion_sort(OctEntry *first, OctEntry *last) {
    if (first == last) return;
    for (OctEntry *i = first + 1; i != last; ++i) {
        if (lt(*i, *first)) {
            OctEntry val = *i;
            for (OctEntry *p = i; p != first; --p) *p = *(p - 1);
            *first = val;
        } else {
            unguarded_linear_insert(i);
        }
    }
}
RUMI_HD void move_median_to_first(OctEntry *result, OctEntry *a, OctEntry *b, OctEntry *c) {
    if (lt(*a, *b)) {
        if (lt(*b, *c)) swp(*result, *b);
        else if (lt(*a, *c)) swp(*result, *c);
        else swp(*result, *a);
    } else if (lt(*a, *c)) swp(*result, *a);
    else if (lt(*b, *c)) swp(*result, *c);
    else swp(*result, *b);
}
RUMI_HD OctEntry *unguarded_partition(OctEntry *first, OctEntry *last, OctEntry *pivot) {
    while (true) {
        while (lt(*first, *pivot)) ++first;
        --last;
        while (lt(*pivot, *last)) --last;
        if (!(first < last)) return first;
        swp(*first, *last);
        ++first;
    }
}
// std::__adjust_heap + std::__push_heap (max-heap on lt)
RUMI_HD void adjust_heap(OctEntry *first, int hole, int len, OctEntry value) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (lt(first[child], first[child - 1])) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && lt(first[parent], value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}
// std::__partial_sort(first, last, last): make_heap + sort_heap (heap_select's scan is empty)
RUMI_HD void heap_sort(OctEntry *first, OctEntry *last) {
    const int len = (int)(last - first);
    if (len >= 2) {
        for (int parent = (len - 2) / 2;; parent--) {
            OctEntry v = first[parent];
            adjust_heap(first, parent, len, v);
            if (parent == 0) break;
        }
    }
    while (last - first > 1) {
        --last;
        OctEntry v = *last;
        *last = *first;
        adjust_heap(first, 0, (int)(last - first), v);
    }
}
}  // namespace sortimpl

// std::sort(first, first+n, compareNodes) as libstdc++ executes it.  The recursion of
// __introsort_loop (recurse on the right part, loop on the left) is unrolled with a small stack.
RUMI_HD void sort_like_libstdcxx(OctEntry *a, int n) {
    using namespace sortimpl;
    if (n <= 1) return;
    int lg = 0;
    for (int t = n; t > 1; t >>= 1) lg++;
    struct Frame { int first, last, depth; };
    Frame stack[64];
    int sp = 0;
    stack[sp++] = Frame{0, n, lg * 2};
    while (sp > 0) {
        Frame f = stack[--sp];
        int first = f.first, last = f.last, depth = f.depth;
        while (last - first > 16) {
            if (depth == 0) { heap_sort(a + first, a + last); break; }
            --depth;
            int mid = first + (last - first) / 2;
            move_median_to_first(a + first, a + first + 1, a + mid, a + last - 1);
            int cut = (int)(unguarded_partition(a + first + 1, a + last, a + first) - a);
            // recursive call on [cut,last) happens BEFORE the loop continues on [first,cut): the two
            // ranges are disjoint, so processing order does not change the result; push the right part.
            stack[sp++] = Frame{cut, last, depth};
            last = cut;
        }
    }
    if (n > 16) {
        insertion_sort(a, a + 16);
        for (OctEntry *i = a + 16; i != a + n; ++i) unguarded_linear_insert(i);
    } else {
        insertion_sort(a, a + n);
    }
}

// ---- serial control state ---------------------------------------------------------------------
struct OctState {
    OctNode *nodes;       // node pool [cap]
    uint16_t *freeIds;    // stack of free node ids [cap]
    OctEntry *open;       // children with > 1 key created by the current round [cap]
    OctEntry *prev;       // sort workspace [cap]
    uint16_t *splitIds;   // nodes divided in the current round (freed after the relabel sweep) [cap]
    int cap;
    int nFree, nOpen, nSplit;
    int head, size;       // list head id (kNil when empty) and lNodes.size()
    int N;                // wanted number of nodes
    int phase;            // 0 = coarse passes, 1 = fine (sorted) rounds, 2 = finished
    int nExpand;
    int overflow;         // node pool exhausted (cannot happen with cap >= 2N+16; reported, not hidden)
};

RUMI_HD int oct_alloc(OctState &s) {
    if (s.nFree == 0) { s.overflow = 1; return kNil; }
    return s.freeIds[--s.nFree];
}
RUMI_HD void oct_push_front(OctState &s, int id) {
    OctNode &nd = s.nodes[id];
    nd.prev = kNil;
    nd.next = (uint16_t)s.head;
    if (s.head != kNil) s.nodes[s.head].prev = (uint16_t)id;
    s.head = id;
    s.size++;
}
RUMI_HD void oct_erase(OctState &s, int id) {
    OctNode &nd = s.nodes[id];
    if (nd.prev != kNil) s.nodes[nd.prev].next = nd.next; else s.head = nd.next;
    if (nd.next != kNil) s.nodes[nd.next].prev = nd.prev;
    s.size--;
}

// DivideNode + the push_front block that follows every call to it (ORBextractor.cc:471-522, :603-637).
// Uses the quadrant counts gathered by the last sweep instead of touching keys.
RUMI_HD void oct_divide(OctState &s, int id, bool countExpand) {
    OctNode &p = s.nodes[id];
    const int hx = (p.x1 - p.x0 + 1) >> 1, hy = (p.y1 - p.y0 + 1) >> 1;   // ceil(float(d)/2)
    const int xs[3] = {p.x0, p.x0 + hx, p.x1}, ys[3] = {p.y0, p.y0 + hy, p.y1};
    for (int q = 0; q < 4; q++) {
        p.child[q] = kNil;
        const int c = p.cnt[q];
        if (c == 0) continue;
        int cid = oct_alloc(s);
        if (cid == kNil) return;
        OctNode &ch = s.nodes[cid];
        ch.x0 = (uint16_t)xs[q & 1]; ch.x1 = (uint16_t)xs[(q & 1) + 1];
        ch.y0 = (uint16_t)ys[q >> 1]; ch.y1 = (uint16_t)ys[(q >> 1) + 1];
        ch.n = (uint16_t)c; ch.noMore = c == 1; ch.split = 0;
        ch.cnt[0] = ch.cnt[1] = ch.cnt[2] = ch.cnt[3] = 0;
        ch.child[0] = ch.child[1] = ch.child[2] = ch.child[3] = kNil;
        p.child[q] = (uint16_t)cid;
        oct_push_front(s, cid);
        if (c > 1) {
            if (countExpand) s.nExpand++;
            s.open[s.nOpen++] = OctEntry{((uint32_t)c << 16) | ch.x0, (uint16_t)cid, 0};
        }
    }
    p.split = 1;
    s.splitIds[s.nSplit++] = (uint16_t)id;
}

// One serial round of the while(!bFinish) loop (ORBextractor.cc:587-702).  After it returns with
// phase != 2 the caller must run a key sweep (relabel + count) and then oct_release_split().
RUMI_HD void oct_round(OctState &s) {
    const int prevSize = s.size;
    if (s.phase == 0) {
        s.nExpand = 0; s.nOpen = 0;
        for (int it = s.head; it != kNil;) {
            const int nxt = s.nodes[it].next;
            if (!s.nodes[it].noMore) {
                oct_divide(s, it, true);
                oct_erase(s, it);
            }
            it = nxt;
        }
        if (s.size >= s.N || s.size == prevSize) s.phase = 2;
        else if (s.size + s.nExpand * 3 > s.N) s.phase = 1;
    } else {
        const int nPrev = s.nOpen;
        for (int i = 0; i < nPrev; i++) s.prev[i] = s.open[i];
        s.nOpen = 0;
        sort_like_libstdcxx(s.prev, nPrev);
        for (int j = nPrev - 1; j >= 0; j--) {
            const int id = s.prev[j].id;
            oct_divide(s, id, false);
            oct_erase(s, id);
            if (s.size >= s.N) break;
        }
        if (s.size >= s.N || s.size == prevSize) s.phase = 2;
    }
    if (s.overflow) s.phase = 2;
}

RUMI_HD void oct_release_split(OctState &s) {
    for (int i = 0; i < s.nSplit; i++) {
        s.nodes[s.splitIds[i]].split = 0;
        s.freeIds[s.nFree++] = s.splitIds[i];
    }
    s.nSplit = 0;
}

// Quadrant of a key inside a node about to be divided: kp.x < n1.UR.x, kp.y < n1.BR.y (:501-511).
RUMI_HD int oct_quadrant(const OctNode &p, int x, int y) {
    const int hx = (p.x1 - p.x0 + 1) >> 1, hy = (p.y1 - p.y0 + 1) >> 1;
    return (x < p.x0 + hx ? 0 : 1) + (y < p.y0 + hy ? 0 : 2);
}

// Candidate packing shared by the FAST kernel, the host and the quadtree:
// x (12 bit) | y (12 bit) << 12 | score (8 bit) << 24, coordinates relative to (16,16).
RUMI_HD int cand_x(uint32_t c) { return (int)(c & 0xFFFu); }
RUMI_HD int cand_y(uint32_t c) { return (int)((c >> 12) & 0xFFFu); }
RUMI_HD int cand_score(uint32_t c) { return (int)(c >> 24); }

// Host driver of the same state machine (used by the CPU tests of the
// replayed sort / list logic).  out receives indices into cand in the reference's result order.
inline int octree_host(const uint32_t *cand, int n, int minX, int maxX, int minY, int maxY, int N,
                       std::vector<int> &out) {
    out.clear();
    if (n <= 0) return 0;
    const int nIni = (int)__builtin_roundf((float)(maxX - minX) / (float)(maxY - minY));
    if (nIni <= 0) return -1;
    const float hX = (float)(maxX - minX) / nIni;
    const int cap = 2 * (N > nIni ? N : nIni) + 16 + nIni;
    std::vector<OctNode> nodes(cap);
    std::vector<uint16_t> freeIds(cap), splitIds(cap), owner(n);
    std::vector<OctEntry> open(cap), prev(cap);
    OctState s{nodes.data(), freeIds.data(), open.data(), prev.data(), splitIds.data(), cap,
               0, 0, 0, kNil, 0, N, 0, 0, 0};
    for (int i = cap - 1; i >= 0; i--) s.freeIds[s.nFree++] = (uint16_t)i;
    // roots (:548-566): push_back order, keys assigned by x / hX
    std::vector<int> rootId(nIni);
    int tail = kNil;
    for (int i = 0; i < nIni; i++) {
        int id = oct_alloc(s);
        OctNode &r = s.nodes[id];
        r = OctNode{};
        r.x0 = (uint16_t)(int)(hX * (float)i); r.x1 = (uint16_t)(int)(hX * (float)(i + 1));
        r.y0 = 0; r.y1 = (uint16_t)(maxY - minY);
        r.next = kNil; r.prev = (uint16_t)tail;
        r.child[0] = r.child[1] = r.child[2] = r.child[3] = kNil;
        if (tail != kNil) s.nodes[tail].next = (uint16_t)id; else s.head = id;
        tail = id; s.size++;
        rootId[i] = id;
    }
    for (int i = 0; i < n; i++) {
        int r = (int)((float)cand_x(cand[i]) / hX);
        owner[i] = (uint16_t)rootId[r];
        s.nodes[rootId[r]].n++;
    }
    for (int i = 0; i < nIni; i++) {          // :570-578
        OctNode &r = s.nodes[rootId[i]];
        if (r.n == 1) r.noMore = 1;
        else if (r.n == 0) { oct_erase(s, rootId[i]); s.freeIds[s.nFree++] = (uint16_t)rootId[i]; }
    }
    auto sweep = [&]() {                      // relabel keys of divided nodes, count quadrants of open nodes
        for (int i = 0; i < n; i++) {
            int id = owner[i];
            const int x = cand_x(cand[i]), y = cand_y(cand[i]);
            if (s.nodes[id].split) { id = s.nodes[id].child[oct_quadrant(s.nodes[id], x, y)]; owner[i] = (uint16_t)id; }
            OctNode &nd = s.nodes[id];
            if (!nd.noMore) nd.cnt[oct_quadrant(nd, x, y)]++;
        }
    };
    // initial quadrant counts of the roots
    sweep();
    while (s.phase != 2) {
        oct_round(s);
        // fresh children start with zero counts; nodes left undivided keep theirs
        for (int i = 0; i < n; i++) {
            int id = owner[i];
            if (!s.nodes[id].split) continue;
            const int x = cand_x(cand[i]), y = cand_y(cand[i]);
            id = s.nodes[id].child[oct_quadrant(s.nodes[id], x, y)];
            owner[i] = (uint16_t)id;
            OctNode &nd = s.nodes[id];
            if (!nd.noMore) nd.cnt[oct_quadrant(nd, x, y)]++;
        }
        oct_release_split(s);
    }
    if (s.overflow) return -2;
    // best key per node: largest response, first in candidate order on ties (:706-721)
    for (int it = s.head; it != kNil; it = s.nodes[it].next) s.nodes[it].best = 0;
    for (int i = 0; i < n; i++) {
        uint32_t v = ((uint32_t)cand_score(cand[i]) << 16) | (uint32_t)(0xFFFF - i);
        OctNode &nd = s.nodes[owner[i]];
        if (v > nd.best) nd.best = v;
    }
    for (int it = s.head; it != kNil; it = s.nodes[it].next) out.push_back(0xFFFF - (int)(s.nodes[it].best & 0xFFFF));
    return (int)out.size();
}

}  // namespace rumi
