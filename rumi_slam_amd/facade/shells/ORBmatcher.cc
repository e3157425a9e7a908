// Replaces R/lib_src/ORBmatcher.cc.  The reference's own header (include/cloud_edge_slam_lib/ORBmatcher.h:36-103) stays as it is; every
// member it declares is defined here, with the reference's exact signature, as a forward to the MI355X facade templates (../ORBmatcher.h,
// compiled into a namespace of their own), which marshal to the C ABI of include/rumi_match.h.  Build this file instead of
// lib_src/ORBmatcher.cc with  -DRUMI_HAVE_SOPHUS -DRUMI_HAVE_OPENCV -I<this repository>/include  and link librumi_hip.so.
#include "ORBmatcher.h"          // the REFERENCE's header (found through the reference's include path)

#include "Frame.h"
#include "KeyFrame.h"
#include "MapPoint.h"

#define RUMI_FACADE_NAMESPACE rumi_facade_impl
#include "../ORBmatcher.h"       // the facade templates (this repository), relative to this file so that the two headers cannot be confused

namespace ORB_SLAM3 {

const int ORBmatcher::TH_HIGH = RUMI_TH_HIGH;
const int ORBmatcher::TH_LOW = RUMI_TH_LOW;
const int ORBmatcher::HISTO_LENGTH = RUMI_HISTO_LENGTH;

ORBmatcher::ORBmatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

int ORBmatcher::DescriptorDistance(const cv::Mat &a, const cv::Mat &b) { return rumi_facade_impl::ORBmatcher::DescriptorDistance(a, b); }

#define RUMI_IMPL rumi_facade_impl::ORBmatcher(mfNNratio, mbCheckOrientation)

int ORBmatcher::SearchByProjection(Frame &F, const std::vector<MapPoint *> &vpMapPoints, const float th, const bool bFarPoints, const float thFarPoints) {
    return RUMI_IMPL.SearchByProjection(F, vpMapPoints, th, bFarPoints, thFarPoints);
}
int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono) {
    return RUMI_IMPL.SearchByProjection(CurrentFrame, LastFrame, th, bMono);
}
int ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const std::set<MapPoint *> &sAlreadyFound, const float th, const int ORBdist) {
    return RUMI_IMPL.SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist);
}
int ORBmatcher::SearchByProjection(KeyFrame *pKF, Sophus::Sim3<float> &Scw, const std::vector<MapPoint *> &vpPoints, std::vector<MapPoint *> &vpMatched, int th,
                                   float ratioHamming) {
    return RUMI_IMPL.SearchByProjection(pKF, Scw, vpPoints, vpMatched, th, ratioHamming);
}
int ORBmatcher::SearchByProjection(KeyFrame *pKF, Sophus::Sim3<float> &Scw, const std::vector<MapPoint *> &vpPoints, const std::vector<KeyFrame *> &vpPointsKFs,
                                   std::vector<MapPoint *> &vpMatched, std::vector<KeyFrame *> &vpMatchedKF, int th, float ratioHamming) {
    return RUMI_IMPL.SearchByProjection(pKF, Scw, vpPoints, vpPointsKFs, vpMatched, vpMatchedKF, th, ratioHamming);
}
int ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint *> &vpMapPointMatches) { return RUMI_IMPL.SearchByBoW(pKF, F, vpMapPointMatches); }
int ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12) { return RUMI_IMPL.SearchByBoW(pKF1, pKF2, vpMatches12); }
int ORBmatcher::SearchForInitialization(Frame &F1, Frame &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12, int windowSize) {
    return RUMI_IMPL.SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize);
}
int ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<std::pair<size_t, size_t>> &vMatchedPairs, const bool bOnlyStereo,
                                       const bool bCoarse) {
    return RUMI_IMPL.SearchForTriangulation(pKF1, pKF2, vMatchedPairs, bOnlyStereo, bCoarse);
}
int ORBmatcher::SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12, const Sophus::Sim3f &S12, const float th) {
    return RUMI_IMPL.SearchBySim3(pKF1, pKF2, vpMatches12, S12, th);
}
int ORBmatcher::Fuse(KeyFrame *pKF, const std::vector<MapPoint *> &vpMapPoints, const float th, const bool bRight) { return RUMI_IMPL.Fuse(pKF, vpMapPoints, th, bRight); }
int ORBmatcher::Fuse(KeyFrame *pKF, Sophus::Sim3f &Scw, const std::vector<MapPoint *> &vpPoints, float th, std::vector<MapPoint *> &vpReplacePoint) {
    return RUMI_IMPL.Fuse(pKF, Scw, vpPoints, th, vpReplacePoint);
}
#undef RUMI_IMPL

// the two protected helpers the reference declares (ORBmatcher.cc:191-196, :1795-1826): their work happens on the device now; kept so that the
// class is complete for anything that derives from it
float ORBmatcher::RadiusByViewingCos(const float &viewCos) { return viewCos > 0.998f ? 2.5f : 4.0f; }
void ORBmatcher::ComputeThreeMaxima(std::vector<int> *histo, const int L, int &ind1, int &ind2, int &ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    ind1 = ind2 = ind3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = (int)histo[i].size();
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) ind3 = -1;
}

}  // namespace ORB_SLAM3
