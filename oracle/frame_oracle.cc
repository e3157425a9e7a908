// ORACLE -- TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
//
// Frame::UndistortKeyPoints and Frame::ComputeImageBounds (R/lib_src/Frame.cc:770-826) restated on the CPU.  Both call
//   cv::undistortPoints(mat, mat, K, mDistCoef, cv::Mat(), mK)
// whose arithmetic lives in OpenCV 3.4 (imgproc/src/undistort.cpp, cvUndistortPointsInternal; not in the tree: PARITY UNPINNED against the reference
// binary, restated from the published algorithm): per point, in double,
//   x = (u - cx) * (1 / fx),  y = (v - cy) * (1 / fy);  x0 = x, y0 = y;
//   5 fixed-point iterations (TermCriteria(COUNT, 5, 0.01) of the 6-argument overload):
//       r2 = x x + y y;  icdist = (1 + ((k6 r2 + k5) r2 + k4) r2) / (1 + ((k3 r2 + k2) r2 + k1) r2)        (k4..k6 = 0 here)
//       dX = 2 p1 x y + p2 (r2 + 2 x x);  dY = p1 (r2 + 2 y y) + 2 p2 x y                                 (thin-prism terms 0)
//       x = (x0 - dX) icdist;  y = (y0 - dY) icdist
//   then the new camera matrix P = K:  u' = fx x + 0 y + cx,  v' = fy y + cy,  w' = 1 / (0 x + 0 y + 1);  result (float)(u' w'), (float)(v' w').
// K comes in as floats (mK is CV_32F) and is widened; mDistCoef = (k1, k2, p1, p2[, k3]) floats, widened.
#include <algorithm>
#include <cstdint>

extern "C" void orc_undistort_points(const float *xy, int n, const float *K4, const float *dist5, float *out) {
    const double fx = K4[0], fy = K4[1], cx = K4[2], cy = K4[3], ifx = 1. / fx, ify = 1. / fy;
    const double k1 = dist5[0], k2 = dist5[1], p1 = dist5[2], p2 = dist5[3], k3 = dist5[4];
    for (int i = 0; i < n; i++) {
        double x = xy[2 * i], y = xy[2 * i + 1];
        x = (x - cx) * ifx; y = (y - cy) * ify;
        const double x0 = x, y0 = y;
        for (int j = 0; j < 5; j++) {
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
            const double deltaX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x) + 0 * r2 + 0 * r2 * r2;
            const double deltaY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y + 0 * r2 + 0 * r2 * r2;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        const double xx = fx * x + 0 * y + cx, yy = 0 * x + fy * y + cy, ww = 1. / (0 * x + 0 * y + 1);
        out[2 * i] = (float)(xx * ww); out[2 * i + 1] = (float)(yy * ww);
    }
}

// Frame::ComputeImageBounds (Frame.cc:799-826): the four image corners undistorted; bounds4 = {mnMinX, mnMinY, mnMaxX, mnMaxY}
extern "C" void orc_image_bounds(int w, int h, const float *K4, const float *dist5, float *bounds4) {
    if (dist5[0] == 0.0f) { bounds4[0] = 0; bounds4[1] = 0; bounds4[2] = (float)w; bounds4[3] = (float)h; return; }
    const float c[8] = {0, 0, (float)w, 0, 0, (float)h, (float)w, (float)h};
    float u[8];
    orc_undistort_points(c, 4, K4, dist5, u);
    bounds4[0] = std::min(u[0], u[4]); bounds4[2] = std::max(u[2], u[6]);
    bounds4[1] = std::min(u[1], u[3]); bounds4[3] = std::max(u[5], u[7]);
}
