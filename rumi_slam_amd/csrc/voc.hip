// Bag-of-words transform on gfx950 (include/rumi_voc.h): DBoW2's vocabulary-tree descent for all features of a frame (or of
// a batch of frames) at once; the ordered-map assembly stays on the host, in the reference's feature order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "rumi_common.h"
#include "rumi_voc.h"

namespace rumi {

struct VocDev {
    const int32_t *childOff;    // [nNodes + 1] CSR over childIds
    const uint32_t *childIds;   // children of every node in increasing node id (= vector<NodeId> children, file order)
    const uint8_t *desc;        // [nNodes][32]
    const double *weight;       // [nNodes]
    const uint32_t *wordId;     // [nNodes] (leaves)
    const uint2 *range;         // [nNodes] (first child, number of children) when every node's children have consecutive ids (DBoW2 creates a node's
};                              // children in one loop, TemplatedVocabulary.h:615-640, so every file it wrote is of that kind); nullptr otherwise

// 16 lanes per feature, 4 features per wave.  Every level: each lane takes children j, j+16, ... of the current node, the
// group keeps the minimum of (distance << 20 | child position) — FORB::distance is an int and `d < best_d` is strict, so the
// earliest child of the smallest distance wins (TemplatedVocabulary.h:1236-1248).
__global__ __launch_bounds__(256) void k_voc_descend(VocDev V, const uint8_t *__restrict__ desc, const int32_t *__restrict__ counts,
                                                     int cap, int nTotal, int nidLevel, uint32_t *__restrict__ wordOut,
                                                     double *__restrict__ weightOut, uint32_t *__restrict__ nodeOut) {
    const int g = blockIdx.x * 16 + (threadIdx.x >> 4), j = threadIdx.x & 15;
    if (g >= nTotal) return;
    if (counts) {                                   // batched form: slot g = frame * cap + i, live iff i < counts[frame][0]
        const int frame = g / cap, i = g - frame * cap;
        if (i >= counts[2 * frame]) return;
    }
    uint32_t q[8];
    const uint32_t *qs = reinterpret_cast<const uint32_t *>(desc + (size_t)g * 32);
#pragma unroll
    for (int k = 0; k < 8; k++) q[k] = qs[k];
    uint32_t node = 0, nid = 0;
    int level = 0;
    while (true) {
        const int c0 = V.childOff[node], c1 = V.childOff[node + 1];
        if (c0 == c1) break;                        // isLeaf()
        uint32_t best = 0xFFFFFFFFu;
        for (int c = c0 + j; c < c1; c += 16) {
            const uint32_t *d = reinterpret_cast<const uint32_t *>(V.desc + (size_t)V.childIds[c] * 32);
            int dist = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) dist += __popc(q[k] ^ d[k]);
            best = min(best, ((uint32_t)dist << 20) | (uint32_t)(c - c0));
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, o, 16));
        node = V.childIds[c0 + (int)(best & 0xFFFFFu)];
        if (++level == nidLevel) nid = node;
    }
    if (j == 0) {
        wordOut[g] = V.wordId[node];
        weightOut[g] = V.weight[node];
        nodeOut[g] = nid;
    }
}

// The same descent for trees whose children lists are id ranges (ORBvoc.txt: k = 10, L = 6, 1.1 M nodes, a 35 MB descriptor table that no cache
// level of one XCD holds): ONE dependent memory round trip per level instead of three.  The lane that scores a child also fetches the child's own
// (first child, count) record -- the same dependency level as its descriptor -- and the winner's record is handed to the group by a lane read, so the
// next level's descriptor addresses are known as soon as the arg-min is.  The descriptor comes in as two 16-byte loads.
__global__ __launch_bounds__(256) void k_voc_descend_ranges(VocDev V, const uint8_t *__restrict__ desc, const int32_t *__restrict__ counts,
                                                            int cap, int nTotal, int nidLevel, uint32_t *__restrict__ wordOut,
                                                            double *__restrict__ weightOut, uint32_t *__restrict__ nodeOut) {
    const int g = blockIdx.x * 16 + (threadIdx.x >> 4), j = threadIdx.x & 15;
    if (g >= nTotal) return;
    if (counts) {
        const int frame = g / cap, i = g - frame * cap;
        if (i >= counts[2 * frame]) return;
    }
    const uint4 *qs = reinterpret_cast<const uint4 *>(desc + (size_t)g * 32);
    const uint4 q0 = qs[0], q1 = qs[1];
    uint2 cur = V.range[0];
    uint32_t node = 0, nid = 0;
    int level = 0;
    const int rowBase = (threadIdx.x & 63) & ~15;
    while (cur.y != 0) {
        uint32_t best = 0xFFFFFFFFu;
        uint2 bestRange = make_uint2(0, 0);
        for (uint32_t c = j; c < cur.y; c += 16) {
            const uint32_t child = cur.x + c;
            const uint4 *d = reinterpret_cast<const uint4 *>(V.desc + (size_t)child * 32);
            const uint4 d0 = d[0], d1 = d[1];
            const uint2 r = V.range[child];
            const int dist = __popc(q0.x ^ d0.x) + __popc(q0.y ^ d0.y) + __popc(q0.z ^ d0.z) + __popc(q0.w ^ d0.w) + __popc(q1.x ^ d1.x) + __popc(q1.y ^ d1.y) +
                             __popc(q1.z ^ d1.z) + __popc(q1.w ^ d1.w);
            const uint32_t key = ((uint32_t)dist << 20) | c;
            if (key < best) { best = key; bestRange = r; }
        }
        uint32_t m = best;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, o, 16));
        const uint32_t c = m & 0xFFFFFu;
        const int src = rowBase + (int)(c & 15);             // the lane that scored the winner (it kept the winner's record: keys are distinct)
        node = cur.x + c;
        cur.x = (uint32_t)__shfl((int)bestRange.x, src); cur.y = (uint32_t)__shfl((int)bestRange.y, src);
        if (++level == nidLevel) nid = node;
    }
    if (j == 0) {
        wordOut[g] = V.wordId[node];
        weightOut[g] = V.weight[node];
        nodeOut[g] = nid;
    }
}

}  // namespace rumi

using namespace rumi;

struct RumiVocabulary {
    int device = 0, nNodes = 0, nWords = 0, L = 0, weighting = 0, scoring = 0;
    int32_t *dChildOff = nullptr; uint32_t *dChildIds = nullptr; uint8_t *dDesc = nullptr; double *dWeight = nullptr; uint32_t *dWordId = nullptr;
    uint2 *dRange = nullptr;         // children as id ranges, when the tree allows it (k_voc_descend_ranges)
    // scratch of the host-array entry points
    int cap = 0;
    uint8_t *dQ = nullptr; uint32_t *dWord = nullptr, *dNode = nullptr; double *dW = nullptr;
};

extern "C" void rumi_voc_destroy(RumiVocabulary *v) {
    if (!v) return;
    (void)hipSetDevice(v->device);
    void *p[] = {v->dChildOff, v->dChildIds, v->dDesc, v->dWeight, v->dWordId, v->dQ, v->dWord, v->dNode, v->dW, v->dRange};
    for (void *q : p) if (q) (void)hipFree(q);
    delete v;
}
extern "C" int32_t rumi_voc_words(const RumiVocabulary *v) { return v ? v->nWords : 0; }
extern "C" int32_t rumi_voc_levels(const RumiVocabulary *v) { return v ? v->L : 0; }

template <class T> static int upload(T **d, const std::vector<T> &h) {
    *d = nullptr;
    HIP_TRY(hipMalloc((void **)d, std::max<size_t>(h.size(), 1) * sizeof(T)));
    if (!h.empty()) HIP_TRY(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return RUMI_OK;
}

extern "C" int rumi_voc_create(int32_t n_nodes, const int32_t *parent, const uint8_t *is_leaf, const uint8_t *desc, const double *weight,
                               int32_t weighting, int32_t scoring, int32_t device, RumiVocabulary **out) {
    if (!out) return RUMI_E_INVALID;
    *out = nullptr;
    if (n_nodes < 2 || !parent || !is_leaf || !desc || !weight || weighting < 0 || weighting > 3 || scoring < 0 || scoring > 5) {
        g_lastError = "rumi_voc_create: bad argument";
        return RUMI_E_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        g_lastError = "no HIP device visible: librumi_hip has no CPU fallback";
        return RUMI_E_NO_DEVICE;
    }
    // children lists in file order (m_nodes[pid].children.push_back(nid), TemplatedVocabulary.h:1389-1390), word ids in file order (:1408-1413)
    std::vector<int32_t> off(n_nodes + 1, 0), depth(n_nodes, 0);
    for (int i = 1; i < n_nodes; i++) {
        if (parent[i] < 0 || parent[i] >= i) { g_lastError = "rumi_voc_create: parent id must precede the node (file order)"; return RUMI_E_INVALID; }
        off[parent[i] + 1]++;
    }
    for (int i = 0; i < n_nodes; i++) off[i + 1] += off[i];
    std::vector<uint32_t> ids(n_nodes - 1), wid(n_nodes, 0);
    std::vector<int32_t> fill(off.begin(), off.end() - 1);
    int nWords = 0, L = 0;
    for (int i = 1; i < n_nodes; i++) {
        ids[fill[parent[i]]++] = (uint32_t)i;
        depth[i] = depth[parent[i]] + 1;
        if (is_leaf[i]) { wid[i] = (uint32_t)nWords++; L = std::max(L, depth[i]); }
    }
    for (int i = 1; i < n_nodes; i++)
        if ((off[i + 1] == off[i]) != (is_leaf[i] != 0)) { g_lastError = "rumi_voc_create: is_leaf disagrees with the tree (a leaf has children or an inner node none)"; return RUMI_E_INVALID; }
    RumiVocabulary *v = new RumiVocabulary();
    if (device >= 0) v->device = device; else if (hipGetDevice(&v->device) != hipSuccess) v->device = 0;
    if (hipSetDevice(v->device) != hipSuccess) { delete v; return RUMI_E_NO_DEVICE; }
    v->nNodes = n_nodes; v->nWords = nWords; v->L = L; v->weighting = weighting; v->scoring = scoring;
    std::vector<uint8_t> d(desc, desc + (size_t)n_nodes * 32);
    std::memset(d.data(), 0, 32);                                   // the root has no descriptor
    std::vector<double> w(weight, weight + n_nodes);
    // children as id ranges?  (a lane of the descent then needs one record per child instead of the offset / id / descriptor chain)
    bool consecutive = true;
    std::vector<uint2> range((size_t)n_nodes);
    for (int i = 0; i < n_nodes && consecutive; i++) {
        const int c0 = off[i], c1 = off[i + 1];
        range[i] = make_uint2(c1 > c0 ? ids[c0] : 0u, (uint32_t)(c1 - c0));
        for (int c = c0 + 1; c < c1; c++) if (ids[c] != ids[c - 1] + 1) { consecutive = false; break; }
    }
    static const bool noRanges = std::getenv("RUMI_VOC_NO_RANGES") != nullptr;     // (A/B measurements and tests of the general kernel)
    int rc;
    if (consecutive && !noRanges && (rc = upload(&v->dRange, range)) != RUMI_OK) { rumi_voc_destroy(v); return rc; }
    if ((rc = upload(&v->dChildOff, off)) != RUMI_OK || (rc = upload(&v->dChildIds, ids)) != RUMI_OK || (rc = upload(&v->dDesc, d)) != RUMI_OK ||
        (rc = upload(&v->dWeight, w)) != RUMI_OK || (rc = upload(&v->dWordId, wid)) != RUMI_OK) {
        rumi_voc_destroy(v);
        return rc;
    }
    *out = v;
    return RUMI_OK;
}

extern "C" int rumi_voc_set_levels(RumiVocabulary *v, int32_t L) {
    if (!v || L < 1 || L > 10) return RUMI_E_INVALID;
    v->L = L;
    return RUMI_OK;
}

extern "C" int rumi_voc_load_text(const char *path, int32_t device, RumiVocabulary **out) {
    if (!path || !out) return RUMI_E_INVALID;
    FILE *f = std::fopen(path, "rb");
    if (!f) { g_lastError = "rumi_voc_load_text: cannot open file"; return RUMI_E_INVALID; }
    // ORBvoc.txt is 145 MB of decimal numbers (1.1 M lines of 35): read whole and parsed by hand (the stream extraction of loadFromTextFile,
    // TemplatedVocabulary.h:1338-1425, takes ~10 s on it; 39 M fscanf calls are no better)
    std::fseek(f, 0, SEEK_END);
    const long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<char> buf((size_t)std::max(sz, 0L) + 1);
    const size_t got = sz > 0 ? std::fread(buf.data(), 1, (size_t)sz, f) : 0;
    std::fclose(f);
    buf[got] = 0;
    const char *p = buf.data(), *end = buf.data() + got;
    auto skip = [&]() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) p++; };
    auto read_int = [&](int *v) -> bool {
        skip();
        if (p >= end) return false;
        bool neg = false;
        if (*p == '-') { neg = true; p++; }
        if (p >= end || *p < '0' || *p > '9') return false;
        long x = 0;
        while (p < end && *p >= '0' && *p <= '9') { x = x * 10 + (*p - '0'); p++; }
        *v = (int)(neg ? -x : x);
        return true;
    };
    int k = 0, L = 0, n1 = 0, n2 = 0;
    if (!read_int(&k) || !read_int(&L) || !read_int(&n1) || !read_int(&n2) || k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) {
        g_lastError = "rumi_voc_load_text: not a vocabulary text file";       // the same header test as TemplatedVocabulary.h:1359
        return RUMI_E_INVALID;
    }
    std::vector<int32_t> parent(1, -1);
    std::vector<uint8_t> leaf(1, 0), desc(32, 0);
    std::vector<double> weight(1, 0.0);
    const size_t guess = (size_t)got / 100 + 16;
    parent.reserve(guess); leaf.reserve(guess); desc.reserve(guess * 32); weight.reserve(guess);
    while (true) {
        int pid, isLeaf;
        if (!read_int(&pid) || !read_int(&isLeaf)) break;
        uint8_t d[32];
        bool ok = true;
        for (int i = 0; i < 32; i++) { int b; if (!read_int(&b)) { ok = false; break; } d[i] = (uint8_t)b; }
        if (!ok) break;
        skip();
        char *e2 = nullptr;
        const double w = std::strtod(p, &e2);
        if (e2 == p) break;
        p = e2;
        parent.push_back(pid); leaf.push_back(isLeaf > 0); desc.insert(desc.end(), d, d + 32); weight.push_back(w);
    }
    const int rc = rumi_voc_create((int32_t)parent.size(), parent.data(), leaf.data(), desc.data(), weight.data(), n2, n1, device, out);
    // DBoW2 keeps the HEADER's L as m_L (TemplatedVocabulary.h:1367) and uses it for the FeatureVector level (nid_level = m_L - levelsup, :1229),
    // whatever the depth of the tree that follows: a file whose deepest leaf is shallower than its header says groups features differently
    if (rc == RUMI_OK) (*out)->L = L;
    return rc;
}

static int launch_descend(RumiVocabulary *v, const uint8_t *dDesc, const int32_t *dCounts, int cap, int nTotal, int levelsup, uint32_t *dWord,
                          double *dW, uint32_t *dNode, hipStream_t st) {
    VocDev V{v->dChildOff, v->dChildIds, v->dDesc, v->dWeight, v->dWordId, v->dRange};
    if (v->dRange) hipLaunchKernelGGL(k_voc_descend_ranges, dim3((nTotal + 15) / 16), dim3(256), 0, st, V, dDesc, dCounts, cap, nTotal, v->L - levelsup, dWord, dW, dNode);
    else hipLaunchKernelGGL(k_voc_descend, dim3((nTotal + 15) / 16), dim3(256), 0, st, V, dDesc, dCounts, cap, nTotal, v->L - levelsup, dWord, dW, dNode);
    HIP_TRY(hipGetLastError());
    return RUMI_OK;
}

extern "C" int rumi_voc_transform_features(RumiVocabulary *v, const uint8_t *desc, int32_t n, int32_t levelsup, uint32_t *word_id, double *weight,
                                           uint32_t *node_id) {
    if (!v || n < 0 || (n > 0 && (!desc || !word_id || !weight || !node_id))) return RUMI_E_INVALID;
    if (n == 0) return RUMI_OK;
    HIP_TRY(hipSetDevice(v->device));
    if (n > v->cap) {
        for (void *p : {(void *)v->dQ, (void *)v->dWord, (void *)v->dNode, (void *)v->dW}) if (p) (void)hipFree(p);
        v->dQ = nullptr; v->dWord = nullptr; v->dNode = nullptr; v->dW = nullptr; v->cap = 0;
        const size_t c = (size_t)n * 2;
        HIP_TRY(hipMalloc((void **)&v->dQ, c * 32)); HIP_TRY(hipMalloc((void **)&v->dWord, c * 4));
        HIP_TRY(hipMalloc((void **)&v->dNode, c * 4)); HIP_TRY(hipMalloc((void **)&v->dW, c * 8));
        v->cap = (int)c;
    }
    HIP_TRY(hipMemcpyAsync(v->dQ, desc, (size_t)n * 32, hipMemcpyHostToDevice, nullptr));
    const int rc = launch_descend(v, v->dQ, nullptr, 0, n, levelsup, v->dWord, v->dW, v->dNode, nullptr);
    if (rc != RUMI_OK) return rc;
    HIP_TRY(hipMemcpy(word_id, v->dWord, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(weight, v->dW, (size_t)n * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(node_id, v->dNode, (size_t)n * 4, hipMemcpyDeviceToHost));
    return RUMI_OK;
}

extern "C" int rumi_voc_transform_batch_device(RumiVocabulary *v, const void *d_desc, const void *d_counts, int32_t nframes, int32_t cap,
                                               int32_t levelsup, void *d_word, void *d_weight, void *d_node, void *hip_stream) {
    if (!v || !d_desc || !d_counts || !d_word || !d_weight || !d_node || nframes < 1 || cap < 1) return RUMI_E_INVALID;
    HIP_TRY(hipSetDevice(v->device));
    return launch_descend(v, (const uint8_t *)d_desc, (const int32_t *)d_counts, cap, nframes * cap, levelsup, (uint32_t *)d_word, (double *)d_weight,
                          (uint32_t *)d_node, (hipStream_t)hip_stream);
}

extern "C" int rumi_voc_transform(RumiVocabulary *v, const uint8_t *desc, int32_t n, int32_t levelsup, uint32_t *bow_ids, double *bow_vals,
                                  int32_t *n_words_out, uint32_t *fv_nodes, int32_t *fv_offsets, uint32_t *fv_indices, int32_t *n_nodes_out) {
    if (!v || n < 0 || !n_words_out || !n_nodes_out || !fv_offsets) return RUMI_E_INVALID;
    *n_words_out = 0; *n_nodes_out = 0; fv_offsets[0] = 0;
    if (n == 0) return RUMI_OK;
    if (!bow_ids || !bow_vals || !fv_nodes || !fv_indices) return RUMI_E_INVALID;
    std::vector<uint32_t> word(n), node(n);
    std::vector<double> w(n);
    const int rc = rumi_voc_transform_features(v, desc, n, levelsup, word.data(), w.data(), node.data());
    if (rc != RUMI_OK) return rc;
    return rumi_voc_assemble(v, n, word.data(), w.data(), node.data(), bow_ids, bow_vals, n_words_out, fv_nodes, fv_offsets, fv_indices, n_nodes_out);
}

extern "C" int rumi_voc_assemble(const RumiVocabulary *v, int32_t n, const uint32_t *word, const double *w, const uint32_t *node, uint32_t *bow_ids,
                                 double *bow_vals, int32_t *n_words_out, uint32_t *fv_nodes, int32_t *fv_offsets, uint32_t *fv_indices, int32_t *n_nodes_out) {
    if (!v || n < 0 || !n_words_out || !n_nodes_out || !fv_offsets) return RUMI_E_INVALID;
    *n_words_out = 0; *n_nodes_out = 0; fv_offsets[0] = 0;
    if (n == 0) return RUMI_OK;
    if (!word || !w || !node || !bow_ids || !bow_vals || !fv_nodes || !fv_indices) return RUMI_E_INVALID;
    // the two ordered maps, filled in feature order (TemplatedVocabulary.h:1147-1190)
    std::map<uint32_t, double> bow;
    std::map<uint32_t, std::vector<uint32_t>> fv;
    const bool tf = v->weighting == 0 || v->weighting == 1;                 // TF_IDF or TF: addWeight; IDF / BINARY: addIfNotExist
    for (int i = 0; i < n; i++) {
        if (!(w[i] > 0)) continue;                                            // stopped word
        auto it = bow.lower_bound(word[i]);
        if (it != bow.end() && it->first == word[i]) { if (tf) it->second += w[i]; }
        else bow.insert(it, std::make_pair(word[i], w[i]));
        fv[node[i]].push_back((uint32_t)i);
    }
    const bool must = v->scoring != 5;                                        // DotProductScoring does not normalise
    const bool l2 = v->scoring == 1;
    if (tf && !bow.empty() && !must) {
        const double nd = (double)bow.size();
        for (auto &kv : bow) kv.second /= nd;
    }
    if (must) {                                                               // BowVector::normalize
        double norm = 0.0;
        if (!l2) { for (auto &kv : bow) norm += std::fabs(kv.second); }
        else { for (auto &kv : bow) norm += kv.second * kv.second; norm = std::sqrt(norm); }
        if (norm > 0.0) for (auto &kv : bow) kv.second /= norm;
    }
    int k = 0;
    for (auto &kv : bow) { bow_ids[k] = kv.first; bow_vals[k] = kv.second; k++; }
    *n_words_out = k;
    int a = 0, pos = 0;
    for (auto &kv : fv) {
        fv_nodes[a] = kv.first;
        for (uint32_t idx : kv.second) fv_indices[pos++] = idx;
        fv_offsets[++a] = pos;
    }
    *n_nodes_out = a;
    return RUMI_OK;
}
