// Facade for the one call the hot path makes on the vocabulary (R/lib_src/Frame.cc:763-768, KeyFrame.cc:245-252):
//     mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4);
// Class surface: DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB> (R/Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:33-290),
// typedef'd to ORBVocabulary in R/include/cloud_edge_slam_lib/ORBVocabulary.h.  The tree descent runs on the GPU
// (include/rumi_voc.h); BowVectorT / FeatureVectorT are DBoW2::BowVector / DBoW2::FeatureVector (std::map subclasses), filled
// in ascending key order, which is what their own insertion would produce.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rumi_voc.h"
#include "rumi_status.h"
#if defined(RUMI_HAVE_OPENCV)
#include <opencv2/core/core.hpp>
#else
#include "cv_shim.h"
#endif

namespace rumi_facade {

class ORBVocabulary {
public:
    ORBVocabulary() = default;
    ~ORBVocabulary() { if (h_) rumi_voc_destroy(h_); }
    ORBVocabulary(const ORBVocabulary &) = delete;
    ORBVocabulary &operator=(const ORBVocabulary &) = delete;

    bool loadFromTextFile(const std::string &filename) {
        if (h_) { rumi_voc_destroy(h_); h_ = nullptr; }
        const int rc = rumi_voc_load_text(filename.c_str(), -1, &h_);
        if (rc != RUMI_OK) { rumi_facade::report("ORBVocabulary::loadFromTextFile", rc); return false; }   // no CPU fallback: a missing GPU is reported, not hidden
        return true;
    }
    bool empty() const { return h_ == nullptr || rumi_voc_words(h_) == 0; }
    unsigned int size() const { return h_ ? (unsigned int)rumi_voc_words(h_) : 0; }

    template <class BowVectorT, class FeatureVectorT>
    void transform(const std::vector<cv::Mat> &features, BowVectorT &v, FeatureVectorT &fv, int levelsup) const {
        v.clear();
        fv.clear();
        if (empty()) return;
        const int n = (int)features.size();
        std::vector<uint8_t> desc((size_t)n * 32);
        for (int i = 0; i < n; i++) std::memcpy(&desc[(size_t)i * 32], features[i].ptr(0), 32);
        std::vector<uint32_t> bowIds(n > 0 ? n : 1), fvNodes(n > 0 ? n : 1), fvIdx(n > 0 ? n : 1);
        std::vector<double> bowVals(n > 0 ? n : 1);
        std::vector<int32_t> fvOff(n + 1);
        int32_t nWords = 0, nNodes = 0;
        if (rumi_voc_transform(h_, desc.data(), n, levelsup, bowIds.data(), bowVals.data(), &nWords, fvNodes.data(), fvOff.data(), fvIdx.data(),
                               &nNodes) != RUMI_OK) {
            rumi_facade::report("ORBVocabulary::transform", RUMI_E_NO_DEVICE);
            return;                                          // empty BowVector / FeatureVector, as for an empty vocabulary
        }
        for (int k = 0; k < nWords; k++) v.insert(v.end(), typename BowVectorT::value_type(bowIds[k], bowVals[k]));
        for (int a = 0; a < nNodes; a++) {
            auto it = fv.insert(fv.end(), typename FeatureVectorT::value_type(fvNodes[a], typename FeatureVectorT::mapped_type()));
            it->second.assign(fvIdx.begin() + fvOff[a], fvIdx.begin() + fvOff[a + 1]);
        }
    }

    RumiVocabulary *handle() const { return h_; }

private:
    RumiVocabulary *h_ = nullptr;
};

}  // namespace rumi_facade
