// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// Scalar C++17 CPU restatement of the reference's ORB front-end
// (R/ = /root/reference/src/rumi-slam/):
//   R/lib_src/ORBextractor.cc:73-143   IC_Angle, computeOrbDescriptor
//   R/lib_src/ORBextractor.cc:405-461  constructor tables
//   R/lib_src/ORBextractor.cc:471-724  ExtractorNode::DivideNode, DistributeOctTree
//   R/lib_src/ORBextractor.cc:726-831  ComputeKeyPointsOctTree
//   R/lib_src/ORBextractor.cc:1014-1112 operator(), ComputePyramid
// plus explicit restatements of the OpenCV 3.4 primitives those lines call
// (cv::FAST 9/16 + NMS, cv::resize INTER_LINEAR 8U, BORDER_REFLECT_101,
// cv::GaussianBlur 7x7 sigma 2 fixed-point 8U path, cv::fastAtan2, cvRound).
//
// PARITY STATUS: "parity unpinned" for every OpenCV-owned step — OpenCV is not vendored in the
// reference, is absent from this image, and the reference holds no tests / golden vectors for
// this path.  The in-tree arithmetic (octree, rBRIEF sampling, ordering) is
// pinned by source only.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
// may use this library.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace orc {

// Same 28-byte POD layout as cv::KeyPoint (pt.x, pt.y, size, angle, response, octave, class_id).
struct KeyPoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");

struct Image {
    int w = 0, h = 0;
    std::vector<uint8_t> d;  // row-major, stride == w
    const uint8_t *row(int y) const { return d.data() + (size_t)y * w; }
    uint8_t *row(int y) { return d.data() + (size_t)y * w; }
};

// ---- restated OpenCV primitives (declared for direct unit testing) ----
int cv_round(double v);                                   // round-half-to-even (lrint / cvtsd2si)
float fast_atan2_deg(float y, float x);                   // cv::fastAtan2
void resize_linear_u8(const Image &src, Image &dst, int dw, int dh);   // cv::resize INTER_LINEAR 8UC1
void gaussian_blur_7x7_s2(const Image &src, Image &dst, int variant = 0);  // cv::GaussianBlur(7x7, 2, 2, REFLECT_101); variant: see the .cc
int fast_corner_score(const uint8_t *p, int stride);      // max(A,B)-1, 0-threshold form (see .cc)
// cv::FAST(sub-image, threshold, nonmax=true): appends (x, y, score) relative to the sub-image origin.
void fast_9_16_nms(const uint8_t *img, int stride, int cols, int rows, int threshold,
                   std::vector<KeyPoint> &out);

class OrbExtractor {
public:
    OrbExtractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);

    // Mirrors ORBextractor::operator(): returns monoIndex, or -1 on an empty image.
    int extract(const uint8_t *img, int w, int h, int stride, int lap0, int lap1,
                std::vector<KeyPoint> &kps, std::vector<uint8_t> &desc);

    // stages (public so tests can pin each one)
    void compute_pyramid(const uint8_t *img, int w, int h, int stride);
    void detect_candidates(int level, std::vector<KeyPoint> &cand) const;   // cell loop, coords rel. to (16,16)
    std::vector<KeyPoint> distribute_octree(const std::vector<KeyPoint> &cand, int minX, int maxX,
                                            int minY, int maxY, int N) const;
    float ic_angle(const Image &im, float x, float y) const;
    void orb_descriptor(const Image &blurred, const KeyPoint &kp, uint8_t *desc32) const;

    int nfeatures, nlevels, iniTh, minTh;
    int blurVariant = 0;  // 0 = fixed-point GaussianBlur (OpenCV >= 3.4.2), 1 = sepFilter2D path of 3.4.0 / 3.4.1 (orb_oracle.cc)
    double scaleFactor;  // the reference stores the float ctor argument in a double member
    std::vector<float> scale, invScale, sigma2, invSigma2;
    std::vector<int> featuresPerLevel, umax;

    // intermediates of the last extract()
    std::vector<Image> pyr, blurred;
    std::vector<std::vector<KeyPoint>> cand;   // per level, before the octree (relative to 16,16)
    std::vector<std::vector<KeyPoint>> sel;    // per level, after octree + angle (level coordinates)
};

}  // namespace orc
