// Scalar math shared by the HIP kernels and their host-side unit tests: exact restatements of the
// few libm / OpenCV scalar routines whose bit patterns decide rBRIEF bits.
// Everything here must be compiled with -ffp-contract=off (no FMA contraction).
#pragma once
#include <cfloat>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define RUMI_HD __host__ __device__ inline
#else
#define RUMI_HD inline
#endif

namespace rumi {

// cvRound on a float: round half to even (SSE cvtss2si / lrintf in the default rounding mode).
RUMI_HD int cv_round_f(float v) { return (int)__builtin_rintf(v); }

// cv::fastAtan2(y, x) in degrees — hal::fastAtan32f scalar tail (7th-order odd polynomial, float).
RUMI_HD float fast_atan2_deg(float y, float x) {
    const float kRad2Deg = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * kRad2Deg;
    const float p3 = -0.3258083974640975f * kRad2Deg;
    const float p5 = 0.1555786518463281f * kRad2Deg;
    const float p7 = -0.04432655554792128f * kRad2Deg;
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// sinf / cosf as glibc >= 2.28 computes them (the routine behind the reference's `cos(angle)` on a
// float, ORBextractor.cc:103-104): double-precision range reduction by pi/2 with the 2^24-prescaled
// quotient, degree-7 / degree-8 double polynomials, ONE rounding to float.  Valid for |x| < 120
// (angles here are in [0, 2*pi]).  Checked bit-for-bit against libm in tests/test_math_cpu.py.
namespace sincosf_impl {
struct Tab { double c0, c1, c2, c3, c4, s1, s2, s3; };
RUMI_HD float poly(double x, double x2, bool negCos, int n) {
    const double sg = negCos ? -1.0 : 1.0;
    const double c0 = sg * 0x1p0, c1 = sg * -0x1.ffffffd0c621cp-2, c2c = sg * 0x1.55553e1068f19p-5,
                 c3 = sg * -0x1.6c087e89a359dp-10, c4 = sg * 0x1.99343027bf8c3p-16;
    const double s1c = -0x1.555545995a603p-3, s2c = 0x1.1107605230bc4p-7, s3c = -0x1.994eb3774cf24p-13;
    if ((n & 1) == 0) {
        double x3 = x * x2;
        double s1 = s2c + x2 * s3c;
        double x7 = x3 * x2;
        double s = x + x3 * s1c;
        return (float)(s + x7 * s1);
    } else {
        double x4 = x2 * x2;
        double cc2 = c3 + x2 * c4;
        double cc1 = c0 + x2 * c1;
        double x6 = x4 * x2;
        double c = cc1 + x4 * c2c;
        return (float)(c + x6 * cc2);
    }
}
RUMI_HD uint32_t top12(float f) {
    uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
    u = __float_as_uint(f);
#else
    std::memcpy(&u, &f, 4);
#endif
    return (u >> 20) & 0x7ff;
}
RUMI_HD float eval(float y, int isCos) {
    double x = y;
    if (top12(y) < top12(0x1.921FB6p-1f)) {          // |y| < pi/4
        if (top12(y) < top12(0x1p-12f)) return isCos ? 1.0f : y;
        return poly(x, x * x, false, isCos);
    }
    const double r = x * 0x1.45F306DC9C883p+23;       // 2/pi * 2^24
    const int n = ((int32_t)r + 0x800000) >> 24;
    x = x - n * 0x1.921FB54442D18p0;                  // pi/2
    const double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    return poly(x * s, x * x, (n & 2) != 0, n ^ isCos);
}
}  // namespace sincosf_impl
RUMI_HD float sinf_glibc(float x) { return sincosf_impl::eval(x, 0); }
RUMI_HD float cosf_glibc(float x) { return sincosf_impl::eval(x, 1); }

}  // namespace rumi
