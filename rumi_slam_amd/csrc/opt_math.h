// SE(3) / camera / robust-kernel scalar math shared by the optimiser kernels (double precision), restating
// G/types/se3quat.h (exp :217-257, map :213, operator* :104-110, normalizeRotation :280-285), Eigen's
// Quaternion <-> Matrix3 conversions, Pinhole::project / projectJac (R/lib_src/CameraModels/Pinhole.cpp:35-49,71-81)
// and RobustKernelHuber::robustify (G/core/robust_kernel_impl.cpp:78-91).
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define RUMI_HD __host__ __device__ inline
#else
#define RUMI_HD inline
#endif

namespace rumi {

struct D3 { double x, y, z; };
struct DQuat { double x, y, z, w; };
struct DSE3 { DQuat r; D3 t; };
struct DCam { double fx, fy, cx, cy; };

// ---- reciprocal, reciprocal square root, square root, sine / cosine of the serial sections ----
// Every lane of a pose-optimisation workgroup executes the 6x6 solve and the SE(3) update of each LM trial, alone on its SIMD (about 5 cycles
// per instruction): the IEEE f64 divide and square root are 25-35 instructions each, sin / cos well over 100, and there are ~40 of them per
// trial.  On the device they are replaced by the hardware estimates (v_rcp_f64 / v_rsq_f64, 5e-8) with two Newton steps (3e-16 measured,
// tools/rsq_probe.hip) and by Taylor polynomials for |x| <= 0.5 (LM rotation steps are far smaller; truncation below 1e-19).  The host side
// (pose packing) keeps the libm forms.  Results move by a few 1e-16 relative: the parity bar of these kernels is 1e-4 (tests/test_optimizer_gpu.py).
RUMI_HD double m_rcp(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rcp(d);
    y = __builtin_fma(__builtin_fma(-d, y, 1.0), y, y);
    y = __builtin_fma(__builtin_fma(-d, y, 1.0), y, y);
    return y;
#else
    return 1.0 / d;
#endif
}
RUMI_HD double m_rsqrt(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rsq(d);
    y = y * __builtin_fma(-0.5 * d * y, y, 1.5);
    y = y * __builtin_fma(-0.5 * d * y, y, 1.5);
    return y;
#else
    return 1.0 / std::sqrt(d);
#endif
}
RUMI_HD double m_sqrt(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    return d > 0 ? d * m_rsqrt(d) : (d == 0 ? 0.0 : __builtin_nan(""));
#else
    return std::sqrt(d);
#endif
}
RUMI_HD void m_sincos(double x, double &s, double &c) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (__builtin_fabs(x) <= 0.5) {
        const double z = x * x;
        double ps = -1.0 / 1307674368000.0;                                   // -1/15!
        ps = __builtin_fma(ps, z, 1.0 / 6227020800.0); ps = __builtin_fma(ps, z, -1.0 / 39916800.0); ps = __builtin_fma(ps, z, 1.0 / 362880.0);
        ps = __builtin_fma(ps, z, -1.0 / 5040.0); ps = __builtin_fma(ps, z, 1.0 / 120.0); ps = __builtin_fma(ps, z, -1.0 / 6.0);
        s = __builtin_fma(ps * z, x, x);
        double pc = 1.0 / 20922789888000.0;                                   // 1/16!
        pc = __builtin_fma(pc, z, -1.0 / 87178291200.0); pc = __builtin_fma(pc, z, 1.0 / 479001600.0); pc = __builtin_fma(pc, z, -1.0 / 3628800.0);
        pc = __builtin_fma(pc, z, 1.0 / 40320.0); pc = __builtin_fma(pc, z, -1.0 / 720.0); pc = __builtin_fma(pc, z, 1.0 / 24.0);
        pc = __builtin_fma(pc, z, -0.5);
        c = __builtin_fma(pc, z, 1.0);
        return;
    }
#endif
    s = sin(x); c = cos(x);
}

RUMI_HD D3 d3_cross(D3 a, D3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

RUMI_HD D3 quat_rotate(const DQuat &q, D3 v) {          // Eigen _transformVector
    const D3 qv{q.x, q.y, q.z};
    D3 uv = d3_cross(qv, v);
    uv = {uv.x + uv.x, uv.y + uv.y, uv.z + uv.z};
    const D3 c = d3_cross(qv, uv);
    return {v.x + q.w * uv.x + c.x, v.y + q.w * uv.y + c.y, v.z + q.w * uv.z + c.z};
}
RUMI_HD D3 se3_map(const DSE3 &T, D3 p) {
    const D3 r = quat_rotate(T.r, p);
    return {r.x + T.t.x, r.y + T.t.y, r.z + T.t.z};
}
RUMI_HD void quat_normalize_pos(DQuat &q) {
    if (q.w < 0) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
    const double rn = m_rsqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    q.x *= rn; q.y *= rn; q.z *= rn; q.w *= rn;
}
RUMI_HD DQuat quat_mul(const DQuat &a, const DQuat &b) {
    return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
            a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
RUMI_HD DQuat quat_from_matrix(const double R[3][3]) {
    DQuat q;
    double t = R[0][0] + R[1][1] + R[2][2];
    if (t > 0) {
        const double r = m_rsqrt(t + 1.0);
        q.w = 0.5 * ((t + 1.0) * r);
        t = 0.5 * r;
        q.x = (R[2][1] - R[1][2]) * t; q.y = (R[0][2] - R[2][0]) * t; q.z = (R[1][0] - R[0][1]) * t;
    } else {
        // largest diagonal element decides the branch (written out: no runtime-indexed arrays on the device)
        if (R[0][0] >= R[1][1] && R[0][0] >= R[2][2]) {
            t = m_sqrt(R[0][0] - R[1][1] - R[2][2] + 1.0);
            q.x = 0.5 * t; t = 0.5 * m_rcp(t);
            q.w = (R[2][1] - R[1][2]) * t; q.y = (R[1][0] + R[0][1]) * t; q.z = (R[2][0] + R[0][2]) * t;
        } else if (R[1][1] >= R[2][2]) {
            t = m_sqrt(R[1][1] - R[2][2] - R[0][0] + 1.0);
            q.y = 0.5 * t; t = 0.5 * m_rcp(t);
            q.w = (R[0][2] - R[2][0]) * t; q.z = (R[2][1] + R[1][2]) * t; q.x = (R[0][1] + R[1][0]) * t;
        } else {
            t = m_sqrt(R[2][2] - R[0][0] - R[1][1] + 1.0);
            q.z = 0.5 * t; t = 0.5 * m_rcp(t);
            q.w = (R[1][0] - R[0][1]) * t; q.x = (R[0][2] + R[2][0]) * t; q.y = (R[1][2] + R[2][1]) * t;
        }
    }
    return q;
}
RUMI_HD void quat_to_matrix(const DQuat &q, double R[3][3]) {
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0][0] = 1 - (tyy + tzz); R[0][1] = txy - twz; R[0][2] = txz + twy;
    R[1][0] = txy + twz; R[1][1] = 1 - (txx + tzz); R[1][2] = tyz - twx;
    R[2][0] = txz - twy; R[2][1] = tyz + twx; R[2][2] = 1 - (txx + tyy);
}
// SE3Quat::exp: u = (omega, upsilon)
RUMI_HD DSE3 se3_exp(const double u[6]) {
    const double wx = u[0], wy = u[1], wz = u[2];
    const double theta = m_sqrt(wx * wx + wy * wy + wz * wz);
    const double O[3][3] = {{0, -wz, wy}, {wz, 0, -wx}, {-wy, wx, 0}};
    double O2[3][3], R[3][3], V[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) O2[i][j] = O[i][0] * O[0][j] + O[i][1] * O[1][j] + O[i][2] * O[2][j];
    double a = 1, b = 1, c = 1;
    const bool small = theta < 0.00001;
    if (!small) {
        double sn, cs;
        m_sincos(theta, sn, cs);
        const double it = m_rcp(theta), it2 = it * it;
        a = sn * it; b = (1 - cs) * it2; c = (theta - sn) * (it2 * it);
    }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            const double I = i == j ? 1.0 : 0.0;
            R[i][j] = small ? I + O[i][j] + O2[i][j] : I + a * O[i][j] + b * O2[i][j];
            V[i][j] = small ? R[i][j] : I + b * O[i][j] + c * O2[i][j];
        }
    DSE3 T;
    T.r = quat_from_matrix(R);
    T.t = {V[0][0] * u[3] + V[0][1] * u[4] + V[0][2] * u[5], V[1][0] * u[3] + V[1][1] * u[4] + V[1][2] * u[5],
           V[2][0] * u[3] + V[2][1] * u[4] + V[2][2] * u[5]};
    quat_normalize_pos(T.r);
    return T;
}
RUMI_HD DSE3 se3_mul(const DSE3 &a, const DSE3 &b) {
    DSE3 r;
    const D3 rt = quat_rotate(a.r, b.t);
    r.t = {a.t.x + rt.x, a.t.y + rt.y, a.t.z + rt.z};
    r.r = quat_mul(a.r, b.r);
    quat_normalize_pos(r.r);
    return r;
}
RUMI_HD DSE3 se3_from_float7(const float *T7) {
    DSE3 T{{T7[0], T7[1], T7[2], T7[3]}, {T7[4], T7[5], T7[6]}};
    quat_normalize_pos(T.r);
    return T;
}
RUMI_HD void se3_to_float7(const DSE3 &T, float *o) {     // Sophus::SE3f(q.cast<float>(), t.cast<float>()) normalises the float quaternion
    const float q0 = (float)T.r.x, q1 = (float)T.r.y, q2 = (float)T.r.z, q3 = (float)T.r.w;
    const float n = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
    o[0] = q0 / n; o[1] = q1 / n; o[2] = q2 / n; o[3] = q3 / n;
    o[4] = (float)T.t.x; o[5] = (float)T.t.y; o[6] = (float)T.t.z;
}

RUMI_HD void cam_project(const DCam &c, D3 p, double &u, double &v) { const double iz = m_rcp(p.z); u = c.fx * p.x * iz + c.cx; v = c.fy * p.y * iz + c.cy; }

// rho[0] = rho(e), rho[1] = rho'(e)
RUMI_HD void huber(double e, double delta, double dsqr, double &rho0, double &rho1) {
    if (e <= dsqr) { rho0 = e; rho1 = 1.; }
    else { const double rs = m_rsqrt(e), s = e * rs; rho0 = 2 * s * delta - dsqr; rho1 = delta * rs; }
}

// d e / d (pose increment) = -projectJac(Xc) * [ -[Xc]x | I ]      (OptimizableTypes.cpp:47-61)
RUMI_HD void jac_pose(const DCam &c, D3 p, double J0[6], double J1[6]) {
    const double iz = m_rcp(p.z), iz2 = iz * iz;
    const double j00 = c.fx * iz, j02 = -c.fx * p.x * iz2, j11 = c.fy * iz, j12 = -c.fy * p.y * iz2;
    // rows of SE3deriv: [0 z -y 1 0 0; -z 0 x 0 1 0; y -x 0 0 0 1]
    J0[0] = -(j02 * p.y);            J0[1] = -(j00 * p.z - j02 * p.x); J0[2] = -(-j00 * p.y);
    J0[3] = -j00;                    J0[4] = 0;                         J0[5] = -j02;
    J1[0] = -(-j11 * p.z + j12 * p.y); J1[1] = -(-j12 * p.x);            J1[2] = -(j11 * p.x);
    J1[3] = 0;                       J1[4] = -j11;                      J1[5] = -j12;
}

// ---- g2o::Sim3 (G/types/sim3.h): rotation quaternion (never re-normalised), translation, scale ----
struct DSim3 { DQuat r; D3 t; double s; };

RUMI_HD DSim3 sim3_from8(const double *S) { return DSim3{{S[0], S[1], S[2], S[3]}, {S[4], S[5], S[6]}, S[7]}; }
RUMI_HD void sim3_to8(const DSim3 &S, double *o) { o[0] = S.r.x; o[1] = S.r.y; o[2] = S.r.z; o[3] = S.r.w; o[4] = S.t.x; o[5] = S.t.y; o[6] = S.t.z; o[7] = S.s; }
RUMI_HD D3 sim3_map(const DSim3 &S, D3 p) {                                    // sim3.h:144-146   s*(r*xyz) + t
    const D3 r = quat_rotate(S.r, p);
    return {S.s * r.x + S.t.x, S.s * r.y + S.t.y, S.s * r.z + S.t.z};
}
RUMI_HD DSim3 sim3_mul(const DSim3 &a, const DSim3 &b) {                       // sim3.h:266-272
    DSim3 o;
    o.r = quat_mul(a.r, b.r);
    const D3 rt = quat_rotate(a.r, b.t);
    o.t = {a.s * rt.x + a.t.x, a.s * rt.y + a.t.y, a.s * rt.z + a.t.z};
    o.s = a.s * b.s;
    return o;
}
RUMI_HD DSim3 sim3_inverse(const DSim3 &a) {                                   // sim3.h:233-236
    const DQuat c{-a.r.x, -a.r.y, -a.r.z, a.r.w};
    const double k = -1. / a.s;
    return DSim3{c, quat_rotate(c, D3{k * a.t.x, k * a.t.y, k * a.t.z}), 1. / a.s};
}
// Sim3(const Vector7d& update): (omega, upsilon, sigma)                          sim3.h:70-142
RUMI_HD DSim3 sim3_exp(const double u[7]) {
    const double wx = u[0], wy = u[1], wz = u[2], sigma = u[6];
    const double theta = m_sqrt(wx * wx + wy * wy + wz * wz);
    const double O[3][3] = {{0, -wz, wy}, {wz, 0, -wx}, {-wy, wx, 0}};
    double O2[3][3], R[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) O2[i][j] = O[i][0] * O[0][j] + O[i][1] * O[1][j] + O[i][2] * O[2][j];
    const double s = exp(sigma), eps = 0.00001;
    const bool smallT = theta < eps;
    double A, B, C;
    if (fabs(sigma) < eps) {
        C = 1;
        if (smallT) { A = 1. / 2.; B = 1. / 6.; }
        else { const double theta2 = theta * theta; A = (1 - cos(theta)) / theta2; B = (theta - sin(theta)) / (theta2 * theta); }
    } else {
        C = (s - 1) / sigma;
        if (smallT) { const double sigma2 = sigma * sigma; A = ((sigma - 1) * s + 1) / sigma2; B = ((0.5 * sigma2 - sigma + 1) * s) / (sigma2 * sigma); }
        else {
            const double a = s * sin(theta), b = s * cos(theta), theta2 = theta * theta, sigma2 = sigma * sigma, c = theta2 + sigma2;
            A = (a * sigma + (1 - b) * theta) / (theta * c);
            B = (C - ((b - 1) * sigma + a * theta) / c) * 1. / theta2;
        }
    }
    const double ra = smallT ? 1.0 : sin(theta) / theta, rb = smallT ? 1.0 : (1 - cos(theta)) / (theta * theta);
    double W[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            const double I = i == j ? 1.0 : 0.0;
            R[i][j] = smallT ? (I + O[i][j]) + O2[i][j] : (I + ra * O[i][j]) + rb * O2[i][j];
            W[i][j] = (A * O[i][j] + B * O2[i][j]) + C * I;
        }
    DSim3 S;
    S.r = quat_from_matrix(R);
    S.t = {W[0][0] * u[3] + W[0][1] * u[4] + W[0][2] * u[5], W[1][0] * u[3] + W[1][1] * u[4] + W[1][2] * u[5],
           W[2][0] * u[3] + W[2][1] * u[4] + W[2][2] * u[5]};
    S.s = s;
    return S;
}

}  // namespace rumi
