// Facade implementation: marshals cv types to the C ABI.  No image processing happens here.
#include "ORBextractor.h"
#include "rumi_status.h"

#include <cassert>
#include <cstdio>
#include <cstdlib>

// Which cv::GaussianBlur the linked OpenCV 3.4 build has (include/rumi_orb.h, RumiOrbConfig.blur_variant): 0 for 3.4.2 and later (fixed-point
// path), 1 for 3.4.0 / 3.4.1.  A maintainer sets it from CV_VERSION at build time, e.g. -DRUMI_FACADE_BLUR_VARIANT=1; with real OpenCV headers it
// is derived below.
#ifndef RUMI_FACADE_BLUR_VARIANT
#if defined(CV_VERSION_MAJOR) && CV_VERSION_MAJOR == 3 && CV_VERSION_MINOR == 4 && defined(CV_VERSION_REVISION) && CV_VERSION_REVISION < 2
#define RUMI_FACADE_BLUR_VARIANT 1
#else
#define RUMI_FACADE_BLUR_VARIANT 0
#endif
#endif

namespace ORB_SLAM3 {

static const int EDGE_THRESHOLD = 19;

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST) {
    RumiOrbConfig cfg{nfeatures, _scaleFactor, nlevels, iniThFAST, minThFAST, 640, 480, 1, -1, 0, RUMI_FACADE_BLUR_VARIANT};
    mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
    mnFeaturesPerLevel.resize(nlevels); umax.resize(16);
    rumi_orb_tables(&cfg, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data(),
                    mnFeaturesPerLevel.data(), umax.data());
    mvImagePyramid.resize(nlevels);
}

ORBextractor::~ORBextractor() { rumi_orb_destroy(handle_); }

RumiOrbConfig ORBextractor::rumiConfig(int width, int height) const {
    return RumiOrbConfig{nfeatures, (float)scaleFactor, nlevels, iniThFAST, minThFAST, width, height, 1, -1, 0, RUMI_FACADE_BLUR_VARIANT};
}

void ORBextractor::ensureHandle(int width, int height) {
    if (handle_ && width <= capW_ && height <= capH_) return;
    rumi_orb_destroy(handle_);
    handle_ = nullptr;
    capW_ = width > capW_ ? width : capW_;
    capH_ = height > capH_ ? height : capH_;
    RumiOrbConfig cfg{nfeatures, (float)scaleFactor, nlevels, iniThFAST, minThFAST, capW_, capH_, 1, -1, 0, RUMI_FACADE_BLUR_VARIANT};
    const int rc = rumi_orb_create(&cfg, &handle_);
    if (rc != RUMI_OK) {                                    // the reference has no error path here: reported (rumi_status.h), operator() then returns -1
        rumi_facade::report("ORBextractor: rumi_orb_create", rc);
        handle_ = nullptr;
    }
}

int ORBextractor::CloudFrameComputeDescriptors(cv::InputArray, const std::vector<cv::KeyPoint> &, cv::OutputArray) { return -1; }

int ORBextractor::operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint> &_keypoints,
                             cv::OutputArray _descriptors, std::vector<int> &vLappingArea) {
    if (_image.empty()) return -1;
    cv::Mat image = _image.getMat();
    assert(image.type() == CV_8UC1);
    ensureHandle(image.cols, image.rows);
    if (!handle_) { _keypoints.clear(); _descriptors.release(); return -1; }

    const int cap = nfeatures + 4 * nlevels + 64;
    static_assert(sizeof(cv::KeyPoint) == sizeof(RumiKeyPoint), "cv::KeyPoint must be the 28-byte POD");
    std::vector<cv::KeyPoint> kps(cap);
    std::vector<uint8_t> desc((size_t)cap * 32);
    int32_t n = 0, mono = -1;
    const int rc = rumi_orb_extract(handle_, image.data, image.cols, image.rows, (int)image.step, vLappingArea[0], vLappingArea[1],
                                    reinterpret_cast<RumiKeyPoint *>(kps.data()), desc.data(), cap, &n, &mono);
    if (rc == RUMI_E_EMPTY) return -1;
    if (rc != RUMI_OK) {                                    // reported (rumi_status.h); the frame gets no key-points -- no CPU fallback
        rumi_facade::report("ORBextractor::operator()", rc);
        _keypoints.clear();
        _descriptors.release();
        return -1;
    }
    kps.resize(n);
    _keypoints = kps;                                       // _keypoints = vector<cv::KeyPoint>(nkeypoints)   (:1044)
    if (n == 0) _descriptors.release();
    else {
        _descriptors.create(n, 32, CV_8U);              // (:1038)
        cv::Mat d = _descriptors.getMat();
        for (int i = 0; i < n; i++) std::memcpy(d.ptr(i), &desc[(size_t)i * 32], 32);
    }
    if (keepPyramid) {
        pyramidStorage_.resize(nlevels);
        for (int l = 0; l < nlevels; l++) {
            int w = 0, h = 0;
            rumi_orb_pyramid_level(handle_, 0, l, 0, EDGE_THRESHOLD, nullptr, 0, &w, &h);
            pyramidStorage_[l].create(h + 2 * EDGE_THRESHOLD, w + 2 * EDGE_THRESHOLD, CV_8U);
            rumi_orb_pyramid_level(handle_, 0, l, 0, EDGE_THRESHOLD, pyramidStorage_[l].data, (int)pyramidStorage_[l].step, &w, &h);
            mvImagePyramid[l] = cv::Mat(h, w, CV_8U, pyramidStorage_[l].ptr(EDGE_THRESHOLD) + EDGE_THRESHOLD, pyramidStorage_[l].step);
        }
    }
    return mono;
}

}  // namespace ORB_SLAM3
