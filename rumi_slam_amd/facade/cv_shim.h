// Minimal stand-ins for the few OpenCV types that cross the ORBextractor boundary, used ONLY when the facade is
// built without OpenCV (this image has none).  In the reference tree define RUMI_HAVE_OPENCV and the real headers are
// used instead; layouts match (cv::KeyPoint is the 28-byte POD RumiKeyPoint mirrors; cv::Mat here is a dense 8-bit or
// 32-bit-float row-major matrix with a step).
#pragma once
#ifdef RUMI_HAVE_OPENCV
#include <opencv2/core/core.hpp>
#else
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0

namespace cv {
struct Point2f { float x = 0, y = 0; };
struct KeyPoint {
    Point2f pt;
    float size = 0, angle = -1, response = 0;
    int octave = 0, class_id = -1;
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");

class Mat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;
    uint8_t *data = nullptr;
    Mat() {}
    Mat(int r, int c, int /*type*/) { create(r, c, CV_8U); }
    Mat(int r, int c, int /*type*/, void *ext, size_t stp) : rows(r), cols(c), step(stp), data((uint8_t *)ext) {}
    void create(int r, int c, int /*type*/) {
        rows = r; cols = c; step = (size_t)c;
        buf_ = std::shared_ptr<uint8_t>(new uint8_t[(size_t)r * c + 1], std::default_delete<uint8_t[]>());
        data = buf_.get();
    }
    void release() { rows = cols = 0; step = 0; data = nullptr; buf_.reset(); }
    bool empty() const { return rows == 0 || cols == 0 || !data; }
    int type() const { return CV_8UC1; }
    uint8_t *ptr(int r) { return data + (size_t)r * step; }
    const uint8_t *ptr(int r) const { return data + (size_t)r * step; }
    Mat getMat() const { return *this; }
private:
    std::shared_ptr<uint8_t> buf_;
};
typedef const Mat &InputArray;
typedef Mat &OutputArray;
}  // namespace cv
#endif
