// ORACLE — TEST INFRASTRUCTURE ONLY (see orb_oracle.h).  CPU restatement of DBoW2's bag-of-words transform as the reference
// calls it from Frame::ComputeBoW (R/lib_src/Frame.cc:763-768):
//   TemplatedVocabulary<FORB::TDescriptor, FORB>::transform(features, BowVector&, FeatureVector&, levelsup)
//       R/Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1126-1199, per-feature descent :1217-1260, text loader :1338-1425
//   FORB::distance                        R/Thirdparty/DBoW2/DBoW2/FORB.cpp:81-101
//   BowVector::addWeight / addIfNotExist / normalize    R/Thirdparty/DBoW2/DBoW2/BowVector.cpp:34-84
//   FeatureVector::addFeature             R/Thirdparty/DBoW2/DBoW2/FeatureVector.cpp:31-45
// DBoW2 is vendored source in the reference tree, so this arithmetic is pinned by source (no third-party library involved).
// The vocabulary file ORBvoc.txt itself is a missing blob: tests build synthetic trees with the same node model.
// One deliberate definition: where DBoW2 leaves *nid unset (a leaf shallower than level L - levelsup) this returns node 0.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <vector>

namespace {

struct Node {
    double weight = 0;
    std::vector<uint32_t> children;
    uint8_t desc[32] = {0};
    uint32_t wordId = 0;
};

struct Voc {
    std::vector<Node> nodes;
    int L = 0, weighting = 0, scoring = 0, nWords = 0;
};

int forb_distance(const uint8_t *a, const uint8_t *b) {           // FORB.cpp:81-101 (SWAR popcount over 8 x int32)
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t x, y;
        std::memcpy(&x, a + 4 * i, 4); std::memcpy(&y, b + 4 * i, 4);
        unsigned int v = x ^ y;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

// TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup)   :1217-1260
void transform_one(const Voc &V, const uint8_t *feature, uint32_t &wordId, double &weight, uint32_t &nid, int levelsup) {
    const int nidLevel = V.L - levelsup;
    nid = 0;
    uint32_t finalId = 0;
    int currentLevel = 0;
    do {
        ++currentLevel;
        const std::vector<uint32_t> &nodes = V.nodes[finalId].children;
        finalId = nodes[0];
        double bestD = forb_distance(feature, V.nodes[finalId].desc);
        for (size_t k = 1; k < nodes.size(); k++) {
            const uint32_t id = nodes[k];
            const double d = forb_distance(feature, V.nodes[id].desc);
            if (d < bestD) { bestD = d; finalId = id; }
        }
        if (currentLevel == nidLevel) nid = finalId;
    } while (!V.nodes[finalId].children.empty());
    wordId = V.nodes[finalId].wordId;
    weight = V.nodes[finalId].weight;
}

}  // namespace

extern "C" {

void *orc_voc_create(int nNodes, const int32_t *parent, const uint8_t *isLeaf, const uint8_t *desc, const double *weight, int weighting,
                     int scoring) {
    Voc *V = new Voc();
    V->nodes.resize(nNodes);
    V->weighting = weighting; V->scoring = scoring;
    std::vector<int> depth(nNodes, 0);
    for (int i = 1; i < nNodes; i++) {                                  // loadFromTextFile :1376-1420
        V->nodes[parent[i]].children.push_back((uint32_t)i);
        std::memcpy(V->nodes[i].desc, desc + (size_t)i * 32, 32);
        V->nodes[i].weight = weight[i];
        depth[i] = depth[parent[i]] + 1;
        if (isLeaf[i]) { V->nodes[i].wordId = (uint32_t)V->nWords++; if (depth[i] > V->L) V->L = depth[i]; }
    }
    return V;
}
void orc_voc_destroy(void *h) { delete static_cast<Voc *>(h); }
void orc_voc_set_levels(void *h, int L) { static_cast<Voc *>(h)->L = L; }   // m_L as a file header gives it (TemplatedVocabulary.h:1367)

void orc_voc_transform_features(void *h, const uint8_t *desc, int n, int levelsup, uint32_t *wordId, double *weight, uint32_t *nodeId) {
    const Voc &V = *static_cast<Voc *>(h);
    for (int i = 0; i < n; i++) transform_one(V, desc + (size_t)i * 32, wordId[i], weight[i], nodeId[i], levelsup);
}

// transform(features, BowVector&, FeatureVector&, levelsup)   :1126-1199
void orc_voc_transform(void *h, const uint8_t *desc, int n, int levelsup, uint32_t *bowIds, double *bowVals, int32_t *nWordsOut,
                       uint32_t *fvNodes, int32_t *fvOffsets, uint32_t *fvIndices, int32_t *nNodesOut) {
    const Voc &V = *static_cast<Voc *>(h);
    std::map<uint32_t, double> v;
    std::map<uint32_t, std::vector<uint32_t>> fv;
    const bool must = V.scoring != 5;                                   // ScoringObject.h:74-89
    const bool l2 = V.scoring == 1;
    if (V.weighting == 0 || V.weighting == 1) {                         // TF_IDF || TF
        for (int i = 0; i < n; i++) {
            uint32_t id, nid; double w;
            transform_one(V, desc + (size_t)i * 32, id, w, nid, levelsup);
            if (w > 0) {
                auto vit = v.lower_bound(id);                           // BowVector::addWeight
                if (vit != v.end() && !(id < vit->first)) vit->second += w;
                else v.insert(vit, std::make_pair(id, w));
                fv[nid].push_back((uint32_t)i);                        // FeatureVector::addFeature
            }
        }
        if (!v.empty() && !must) {
            const double nd = (double)v.size();
            for (auto &kv : v) kv.second /= nd;
        }
    } else {                                                            // IDF || BINARY
        for (int i = 0; i < n; i++) {
            uint32_t id, nid; double w;
            transform_one(V, desc + (size_t)i * 32, id, w, nid, levelsup);
            if (w > 0) {
                auto vit = v.lower_bound(id);                           // BowVector::addIfNotExist
                if (vit == v.end() || id < vit->first) v.insert(vit, std::make_pair(id, w));
                fv[nid].push_back((uint32_t)i);
            }
        }
    }
    if (must) {                                                         // BowVector::normalize
        double norm = 0.0;
        if (!l2) { for (auto &kv : v) norm += std::fabs(kv.second); }
        else { for (auto &kv : v) norm += kv.second * kv.second; norm = std::sqrt(norm); }
        if (norm > 0.0) for (auto &kv : v) kv.second /= norm;
    }
    int k = 0;
    for (auto &kv : v) { bowIds[k] = kv.first; bowVals[k] = kv.second; k++; }
    *nWordsOut = k;
    int a = 0, pos = 0;
    fvOffsets[0] = 0;
    for (auto &kv : fv) {
        fvNodes[a] = kv.first;
        for (uint32_t idx : kv.second) fvIndices[pos++] = idx;
        fvOffsets[++a] = pos;
    }
    *nNodesOut = a;
}

}  // extern "C"
