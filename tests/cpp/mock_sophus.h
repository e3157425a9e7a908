// Minimal stand-ins for the few Eigen / Sophus types the RUMI_HAVE_SOPHUS sections of rumi-slam_amd/facade/*.h touch, so that
// those sections are COMPILED and RUN in this image (which has neither library).  Test scaffolding only: it gives the facade
// code something with the same member names to instantiate against; it is not a model of Eigen's or Sophus's rounding.
#pragma once
#include <cmath>

namespace Eigen {
struct Vector3d {
    double v[3];
    Vector3d() : v{0, 0, 0} {}
    Vector3d(double x, double y, double z) : v{x, y, z} {}
    double operator()(int i) const { return v[i]; }
};
struct Matrix3d { double m[3][3]; double operator()(int r, int c) const { return m[r][c]; } };
struct Quaterniond {
    double qw, qx, qy, qz;
    Quaterniond() : qw(1), qx(0), qy(0), qz(0) {}
    Quaterniond(double w, double x, double y, double z) : qw(w), qx(x), qy(y), qz(z) {}
    double x() const { return qx; } double y() const { return qy; } double z() const { return qz; } double w() const { return qw; }
};
struct Vector2f { float v[2]; float operator()(int i) const { return v[i]; } float &operator()(int i) { return v[i]; } };
struct Vector3f {
    float v[3];
    Vector3f() : v{0, 0, 0} {}
    Vector3f(float x, float y, float z) : v{x, y, z} {}
    float operator()(int i) const { return v[i]; }
    float &operator()(int i) { return v[i]; }
    Vector3f operator/(float s) const { return Vector3f(v[0] / s, v[1] / s, v[2] / s); }
    Vector3f operator*(float s) const { return Vector3f(v[0] * s, v[1] * s, v[2] * s); }
    Vector3f operator+(const Vector3f &o) const { return Vector3f(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
    Vector3f operator-(const Vector3f &o) const { return Vector3f(v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]); }
    template <class T> Vector3d cast() const { return Vector3d((double)v[0], (double)v[1], (double)v[2]); }
};
struct Matrix3f {
    float m[3][3];
    Matrix3f() : m{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}} {}
    float operator()(int r, int c) const { return m[r][c]; }
    float &operator()(int r, int c) { return m[r][c]; }
    template <class T> Matrix3d cast() const { Matrix3d d; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) d.m[r][c] = m[r][c]; return d; }
    Matrix3f transpose() const { Matrix3f t; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) t.m[r][c] = m[c][r]; return t; }
    Matrix3f operator*(const Matrix3f &o) const {
        Matrix3f p;
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) p.m[r][c] = m[r][0] * o.m[0][c] + m[r][1] * o.m[1][c] + m[r][2] * o.m[2][c];
        return p;
    }
    Vector3f operator*(const Vector3f &x) const {
        return Vector3f(m[0][0] * x(0) + m[0][1] * x(1) + m[0][2] * x(2), m[1][0] * x(0) + m[1][1] * x(1) + m[1][2] * x(2),
                        m[2][0] * x(0) + m[2][1] * x(1) + m[2][2] * x(2));
    }
    Matrix3f inverse() const {
        const float a = m[0][0], b = m[0][1], c = m[0][2], d = m[1][0], e = m[1][1], f = m[1][2], g = m[2][0], h = m[2][1], i = m[2][2];
        const float det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g), s = 1.0f / det;
        Matrix3f r;
        r.m[0][0] = (e * i - f * h) * s; r.m[0][1] = (c * h - b * i) * s; r.m[0][2] = (b * f - c * e) * s;
        r.m[1][0] = (f * g - d * i) * s; r.m[1][1] = (a * i - c * g) * s; r.m[1][2] = (c * d - a * f) * s;
        r.m[2][0] = (d * h - e * g) * s; r.m[2][1] = (b * g - a * h) * s; r.m[2][2] = (a * e - b * d) * s;
        return r;
    }
};
struct Quaternionf {
    float qw, qx, qy, qz;
    Quaternionf() : qw(1), qx(0), qy(0), qz(0) {}
    Quaternionf(float w, float x, float y, float z) : qw(w), qx(x), qy(y), qz(z) {}
    explicit Quaternionf(const Matrix3f &R) {
        const float tr = R(0, 0) + R(1, 1) + R(2, 2);
        if (tr > 0) { const float s = std::sqrt(tr + 1.0f) * 2; qw = 0.25f * s; qx = (R(2, 1) - R(1, 2)) / s; qy = (R(0, 2) - R(2, 0)) / s; qz = (R(1, 0) - R(0, 1)) / s; }
        else if (R(0, 0) > R(1, 1) && R(0, 0) > R(2, 2)) { const float s = std::sqrt(1.0f + R(0, 0) - R(1, 1) - R(2, 2)) * 2; qw = (R(2, 1) - R(1, 2)) / s; qx = 0.25f * s; qy = (R(0, 1) + R(1, 0)) / s; qz = (R(0, 2) + R(2, 0)) / s; }
        else if (R(1, 1) > R(2, 2)) { const float s = std::sqrt(1.0f + R(1, 1) - R(0, 0) - R(2, 2)) * 2; qw = (R(0, 2) - R(2, 0)) / s; qx = (R(0, 1) + R(1, 0)) / s; qy = 0.25f * s; qz = (R(1, 2) + R(2, 1)) / s; }
        else { const float s = std::sqrt(1.0f + R(2, 2) - R(0, 0) - R(1, 1)) * 2; qw = (R(1, 0) - R(0, 1)) / s; qx = (R(0, 2) + R(2, 0)) / s; qy = (R(1, 2) + R(2, 1)) / s; qz = 0.25f * s; }
    }
    float x() const { return qx; } float y() const { return qy; } float z() const { return qz; } float w() const { return qw; }
    Matrix3f toRotationMatrix() const {
        Matrix3f R;
        R(0, 0) = 1 - 2 * (qy * qy + qz * qz); R(0, 1) = 2 * (qx * qy - qz * qw); R(0, 2) = 2 * (qx * qz + qy * qw);
        R(1, 0) = 2 * (qx * qy + qz * qw); R(1, 1) = 1 - 2 * (qx * qx + qz * qz); R(1, 2) = 2 * (qy * qz - qx * qw);
        R(2, 0) = 2 * (qx * qz - qy * qw); R(2, 1) = 2 * (qy * qz + qx * qw); R(2, 2) = 1 - 2 * (qx * qx + qy * qy);
        return R;
    }
};
// stand-in for Eigen::Matrix<double, 7, 7> (OptimizeSim3's mAcumHessian)
struct Matrix77d { double m[7][7]; void setZero() { for (auto &r : m) for (double &x : r) x = 0; } };
struct Matrix4f {
    float m[4][4];
    float operator()(int r, int c) const { return m[r][c]; }
    float &operator()(int r, int c) { return m[r][c]; }
    static Matrix4f Identity() { Matrix4f I; for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) I.m[r][c] = r == c ? 1.f : 0.f; return I; }
};
struct Matrix4d { double m[4][4]; };
}  // namespace Eigen

namespace Sophus {
struct SO3f {
    static Eigen::Matrix3f hat(const Eigen::Vector3f &w) {
        Eigen::Matrix3f M;
        M(0, 0) = 0; M(0, 1) = -w(2); M(0, 2) = w(1); M(1, 0) = w(2); M(1, 1) = 0; M(1, 2) = -w(0); M(2, 0) = -w(1); M(2, 1) = w(0); M(2, 2) = 0;
        return M;
    }
};
struct SE3f {
    Eigen::Quaternionf q; Eigen::Vector3f t;
    SE3f() {}
    SE3f(const Eigen::Quaternionf &q_, const Eigen::Vector3f &t_) : q(q_), t(t_) {}
    SE3f(const Eigen::Matrix3f &R, const Eigen::Vector3f &t_) : q(R), t(t_) {}
    Eigen::Quaternionf unit_quaternion() const { return q; }
    Eigen::Vector3f translation() const { return t; }
    Eigen::Matrix3f rotationMatrix() const { return q.toRotationMatrix(); }
    SE3f inverse() const { const Eigen::Matrix3f Rt = rotationMatrix().transpose(); return SE3f(Rt, (Rt * t) * -1.0f); }
    SE3f operator*(const SE3f &o) const { return SE3f(rotationMatrix() * o.rotationMatrix(), rotationMatrix() * o.t + t); }
    Eigen::Vector3f operator*(const Eigen::Vector3f &p) const { return rotationMatrix() * p + t; }
};
template <class S> struct Sim3 {
    Eigen::Matrix3f R; Eigen::Vector3f t; S s = 1;
    Sim3() {}
    Sim3(const Eigen::Matrix3f &R_, const Eigen::Vector3f &t_, S s_) : R(R_), t(t_), s(s_) {}
    Eigen::Matrix3f rotationMatrix() const { return R; }
    Eigen::Vector3f translation() const { return t; }
    S scale() const { return s; }
    Sim3 inverse() const { const Eigen::Matrix3f Rt = R.transpose(); return Sim3(Rt, (Rt * t) * (-1.0f / s), 1.0f / s); }
    Eigen::Vector3f operator*(const Eigen::Vector3f &p) const { return (R * p) * s + t; }
};
using Sim3f = Sim3<float>;
}  // namespace Sophus

// g2o::Sim3 stand-in (double): rotation as a quaternion, map = s * (r * p) + t, composition and inverse as in G/types/sim3.h
namespace g2o {
struct Sim3 {
    Eigen::Quaterniond r{1, 0, 0, 0}; Eigen::Vector3d t{0, 0, 0}; double s = 1;
    Sim3() {}
    Sim3(const Eigen::Quaterniond &r_, const Eigen::Vector3d &t_, double s_) : r(r_), t(t_), s(s_) {}
    Sim3(const Eigen::Matrix3d &R, const Eigen::Vector3d &t_, double s_) : t(t_), s(s_) {
        Eigen::Matrix3f Rf;
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Rf(a, b) = (float)R(a, b);
        const Eigen::Quaternionf q(Rf);
        r = Eigen::Quaterniond{q.w(), q.x(), q.y(), q.z()};
    }
    Eigen::Quaterniond rotation() const { return r; }
    Eigen::Vector3d translation() const { return t; }
    double scale() const { return s; }
    static Eigen::Vector3d rot(const Eigen::Quaterniond &q, const Eigen::Vector3d &v) {
        const double ux = 2 * (q.qy * v.v[2] - q.qz * v.v[1]), uy = 2 * (q.qz * v.v[0] - q.qx * v.v[2]), uz = 2 * (q.qx * v.v[1] - q.qy * v.v[0]);
        return Eigen::Vector3d{v.v[0] + q.qw * ux + (q.qy * uz - q.qz * uy), v.v[1] + q.qw * uy + (q.qz * ux - q.qx * uz), v.v[2] + q.qw * uz + (q.qx * uy - q.qy * ux)};
    }
    Sim3 inverse() const {
        const Eigen::Quaterniond c{r.qw, -r.qx, -r.qy, -r.qz};
        const Eigen::Vector3d rt = rot(c, t);
        return Sim3(c, Eigen::Vector3d{-rt.v[0] / s, -rt.v[1] / s, -rt.v[2] / s}, 1.0 / s);
    }
    Sim3 operator*(const Sim3 &o) const {
        const Eigen::Quaterniond q{r.qw * o.r.qw - r.qx * o.r.qx - r.qy * o.r.qy - r.qz * o.r.qz, r.qw * o.r.qx + r.qx * o.r.qw + r.qy * o.r.qz - r.qz * o.r.qy,
                                   r.qw * o.r.qy + r.qy * o.r.qw + r.qz * o.r.qx - r.qx * o.r.qz, r.qw * o.r.qz + r.qz * o.r.qw + r.qx * o.r.qy - r.qy * o.r.qx};
        const Eigen::Vector3d rt = rot(r, o.t);
        return Sim3(q, Eigen::Vector3d{s * rt.v[0] + t.v[0], s * rt.v[1] + t.v[1], s * rt.v[2] + t.v[2]}, s * o.s);
    }
};
}  // namespace g2o
