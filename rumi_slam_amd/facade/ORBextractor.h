// Drop-in ORB_SLAM3::ORBextractor over the MI355X C ABI (include/rumi_orb.h).
// Same class surface as R/include/cloud_edge_slam_lib/ORBextractor.h:42-111 (constructor, operator(), the six getters,
// the public mvImagePyramid); Tracking / Frame / KFDSample code compiles against it unchanged.
#pragma once
#include <list>
#include <vector>

#include "cv_shim.h"
#include "rumi_orb.h"

namespace ORB_SLAM3 {

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
    ~ORBextractor();
    ORBextractor(const ORBextractor &) = delete;
    ORBextractor &operator=(const ORBextractor &) = delete;

    // Declared by the reference, never called there (zero callers): descriptors of given key-points on the raw
    // image.  Kept for ABI completeness; returns -1 (not implemented on the device path).
    int CloudFrameComputeDescriptors(cv::InputArray _image, const std::vector<cv::KeyPoint> &_keypoints, cv::OutputArray _descriptors);

    // Compute the ORB features and descriptors on an image.  Mask is ignored, as in the reference.
    int operator()(cv::InputArray _image, cv::InputArray _mask, std::vector<cv::KeyPoint> &_keypoints,
                   cv::OutputArray _descriptors, std::vector<int> &vLappingArea);

    int inline GetLevels() { return nlevels; }
    float inline GetScaleFactor() { return scaleFactor; }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // facade-only: the constructor arguments as the C ABI takes them (rumi_facade::TrackFrame creates its tracker from the frame's extractor)
    RumiOrbConfig rumiConfig(int width, int height) const;

    // Filled after every call when keepPyramid is true (stereo matching reads it, Frame.cc:834,918-932); each level is a
    // view with the reference's 19-px BORDER_REFLECT_101 frame around it.  Mono tracking never reads it: set false there.
    std::vector<cv::Mat> mvImagePyramid;
    bool keepPyramid = false;

protected:
    void ensureHandle(int width, int height);

    int nfeatures;
    double scaleFactor;
    int nlevels;
    int iniThFAST;
    int minThFAST;
    std::vector<int> mnFeaturesPerLevel;
    std::vector<int> umax;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;

    RumiOrb *handle_ = nullptr;
    int capW_ = 0, capH_ = 0;
    std::vector<cv::Mat> pyramidStorage_;
};

}  // namespace ORB_SLAM3
