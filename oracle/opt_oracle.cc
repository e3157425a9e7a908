// ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement (double precision, scalar) of the reference's motion-only BA
// and local BA, including the g2o machinery they run on.  R/ = /root/reference/src/rumi-slam/, G/ = R/Thirdparty/g2o/g2o/.
//   Optimizer::PoseOptimization                      R/lib_src/Optimizer.cc:723-1001
//   Optimizer::LocalBundleAdjustment                 R/lib_src/Optimizer.cc:1003-1355   (graph already flattened)
//   EdgeSE3ProjectXYZOnlyPose / EdgeSE3ProjectXYZ    R/include/cloud_edge_slam_lib/OptimizableTypes.h:41-50,98-109, R/lib_src/OptimizableTypes.cpp:47-61,135-156
//   Pinhole::project / projectJac                    R/lib_src/CameraModels/Pinhole.cpp:35-49,71-81
//   OptimizationAlgorithmLevenberg::solve            G/core/optimization_algorithm_levenberg.cpp:61-194
//   SparseOptimizer::optimize / activeRobustChi2     G/core/sparse_optimizer.cpp:100-114,354-419
//   BlockSolver buildSystem/setLambda/solve (Schur)  G/core/block_solver.hpp:353-486,501-604
//   constructQuadraticForm (unary / binary)          G/core/base_unary_edge.hpp:42-72, G/core/base_binary_edge.hpp:54-120
//   RobustKernelHuber::robustify, robustInformation  G/core/robust_kernel_impl.cpp:78-91, G/core/base_edge.h:96-102
//   SE3Quat exp / map / operator* / normalize         G/types/se3quat.h:104-121,217-285; VertexSE3Expmap::oplusImpl types_six_dof_expmap.h:73-76
// The linear solvers (Eigen LDLT 6x6; SimplicialLDLT on the reduced pose system) are replaced by a dense Cholesky:
// Eigen is not in the tree and the target is 1e-4 relative on poses / landmarks, not bit-exactness.
// PARITY STATUS: "parity unpinned" against the reference binary (no tests / golden vectors exist; g2o cannot be
// built here without Eigen); pinned by source and by the properties in tests/test_optimizer_cpu.py.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

namespace orcopt {

struct V3 { double x, y, z; };
struct Quat { double x, y, z, w; };
struct SE3 { Quat r; V3 t; };

static inline V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// Eigen QuaternionBase::_transformVector: v + w*uv + q.vec x uv, uv = 2 q.vec x v
static inline V3 rotate(const Quat &q, V3 v) {
    V3 qv{q.x, q.y, q.z};
    V3 uv = cross(qv, v);
    uv = {uv.x + uv.x, uv.y + uv.y, uv.z + uv.z};
    V3 c = cross(qv, uv);
    return {v.x + q.w * uv.x + c.x, v.y + q.w * uv.y + c.y, v.z + q.w * uv.z + c.z};
}
static inline V3 se3_map(const SE3 &T, V3 p) { return add(rotate(T.r, p), T.t); }      // SE3Quat::map

static inline void normalize_rotation(Quat &q) {                                        // se3quat.h:280-285
    if (q.w < 0) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
    const double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    q.x /= n; q.y /= n; q.z /= n; q.w /= n;
}
static inline Quat qmul(const Quat &a, const Quat &b) {
    return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
            a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
// Eigen Quaternion(Matrix3) constructor
static Quat quat_from_R(const double R[3][3]) {
    Quat q;
    double t = R[0][0] + R[1][1] + R[2][2];
    if (t > 0) {
        t = std::sqrt(t + 1.0);
        q.w = 0.5 * t;
        t = 0.5 / t;
        q.x = (R[2][1] - R[1][2]) * t; q.y = (R[0][2] - R[2][0]) * t; q.z = (R[1][0] - R[0][1]) * t;
    } else {
        int i = 0;
        if (R[1][1] > R[0][0]) i = 1;
        if (R[2][2] > R[i][i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(R[i][i] - R[j][j] - R[k][k] + 1.0);
        double v[3];
        v[i] = 0.5 * t;
        t = 0.5 / t;
        q.w = (R[k][j] - R[j][k]) * t;
        v[j] = (R[j][i] + R[i][j]) * t;
        v[k] = (R[k][i] + R[i][k]) * t;
        q.x = v[0]; q.y = v[1]; q.z = v[2];
    }
    return q;
}
static void quat_to_R(const Quat &q, double R[3][3]) {                                   // Eigen toRotationMatrix
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0][0] = 1 - (tyy + tzz); R[0][1] = txy - twz; R[0][2] = txz + twy;
    R[1][0] = txy + twz; R[1][1] = 1 - (txx + tzz); R[1][2] = tyz - twx;
    R[2][0] = txz - twy; R[2][1] = tyz + twx; R[2][2] = 1 - (txx + tyy);
}
// SE3Quat::exp(update): first three = rotation, last three = translation   se3quat.h:217-257
static SE3 se3_exp(const double u[6]) {
    const double wx = u[0], wy = u[1], wz = u[2];
    const double theta = std::sqrt(wx * wx + wy * wy + wz * wz);
    const double O[3][3] = {{0, -wz, wy}, {wz, 0, -wx}, {-wy, wx, 0}};
    double O2[3][3], R[3][3], V[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { O2[i][j] = 0; for (int k = 0; k < 3; k++) O2[i][j] += O[i][k] * O[k][j]; }
    if (theta < 0.00001) {
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { R[i][j] = (i == j) + O[i][j] + O2[i][j]; V[i][j] = R[i][j]; }
    } else {
        const double a = std::sin(theta) / theta, b = (1 - std::cos(theta)) / (theta * theta), c = (theta - std::sin(theta)) / std::pow(theta, 3);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { R[i][j] = (i == j) + a * O[i][j] + b * O2[i][j]; V[i][j] = (i == j) + b * O[i][j] + c * O2[i][j]; }
    }
    SE3 T;
    T.r = quat_from_R(R);
    T.t = {V[0][0] * u[3] + V[0][1] * u[4] + V[0][2] * u[5], V[1][0] * u[3] + V[1][1] * u[4] + V[1][2] * u[5],
           V[2][0] * u[3] + V[2][1] * u[4] + V[2][2] * u[5]};
    normalize_rotation(T.r);                      // SE3Quat(q, t) constructor
    return T;
}
static SE3 se3_mul(const SE3 &a, const SE3 &b) {  // SE3Quat::operator*
    SE3 r = a;
    r.t = add(r.t, rotate(a.r, b.t));
    r.r = qmul(a.r, b.r);
    normalize_rotation(r.r);
    return r;
}
static SE3 se3_from_float7(const float *T7) {     // g2o::SE3Quat(Tcw.unit_quaternion().cast<double>(), Tcw.translation().cast<double>())
    SE3 T{{T7[0], T7[1], T7[2], T7[3]}, {T7[4], T7[5], T7[6]}};
    normalize_rotation(T.r);
    return T;
}
static void se3_to_float7(const SE3 &T, float *o) {   // Sophus::SE3f(rotation().cast<float>(), translation().cast<float>()) normalises the float quaternion
    float q[4] = {(float)T.r.x, (float)T.r.y, (float)T.r.z, (float)T.r.w};
    const float n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; i++) o[i] = q[i] / n;
    o[4] = (float)T.t.x; o[5] = (float)T.t.y; o[6] = (float)T.t.z;
}

struct Cam { double fx, fy, cx, cy; };
static inline void project(const Cam &c, V3 p, double &u, double &v) { u = c.fx * p.x / p.z + c.cx; v = c.fy * p.y / p.z + c.cy; }
static inline void huber(double e, double delta, double dsqr, double rho[3]) {
    if (e <= dsqr) { rho[0] = e; rho[1] = 1.; rho[2] = 0.; }
    else { const double s = std::sqrt(e); rho[0] = 2 * s * delta - dsqr; rho[1] = delta / s; rho[2] = -0.5 * rho[1] / e; }
}

// dense Cholesky solve of a symmetric system; false when a pivot is not positive (g2o: !isPositive / info != Success)
static bool chol_solve(std::vector<double> A, int n, const double *b, double *x) {
    for (int j = 0; j < n; j++) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; k++) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
        if (!(d > 0) || !std::isfinite(d)) return false;
        d = std::sqrt(d);
        A[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[(size_t)i * n + j];
            for (int k = 0; k < j; k++) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
            A[(size_t)i * n + j] = s / d;
        }
    }
    std::vector<double> y(n);
    for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= A[(size_t)i * n + k] * y[k]; y[i] = s / A[(size_t)i * n + i]; }
    for (int i = n - 1; i >= 0; i--) { double s = y[i]; for (int k = i + 1; k < n; k++) s -= A[(size_t)k * n + i] * x[k]; x[i] = s / A[(size_t)i * n + i]; }
    return true;
}

// Jacobian of the reprojection error w.r.t. the pose increment: -projectJac(Xc) * [ -[Xc]x | I ]   (2 x 6)
static inline void jac_pose(const Cam &c, V3 p, double J[2][6]) {
    const double x = p.x, y = p.y, z = p.z;
    const double j00 = c.fx / z, j02 = -c.fx * x / (z * z), j11 = c.fy / z, j12 = -c.fy * y / (z * z);
    const double D[3][6] = {{0, z, -y, 1, 0, 0}, {-z, 0, x, 0, 1, 0}, {y, -x, 0, 0, 0, 1}};
    for (int k = 0; k < 6; k++) { J[0][k] = -(j00 * D[0][k] + j02 * D[2][k]); J[1][k] = -(j11 * D[1][k] + j12 * D[2][k]); }
}

// ---------------------------------------------------------------------------------------------------------------
// One g2o optimize(iterations) run of Levenberg-Marquardt over a generic problem described by callbacks.
// ---------------------------------------------------------------------------------------------------------------
struct LMProblem {
    virtual ~LMProblem() {}
    virtual double robust_chi2() = 0;                 // computeActiveErrors + activeRobustChi2
    virtual void build() = 0;                         // buildSystem at the current estimate
    virtual double max_diag() = 0;                    // computeLambdaInit's max |H_jj|
    virtual bool solve(double lambda) = 0;            // setLambda + solve + restoreDiagonal -> x
    virtual double scale(double lambda) = 0;          // sum_j x_j (lambda x_j + b_j)
    virtual void push() = 0; virtual void pop() = 0;  // save / restore estimates
    virtual void update() = 0;                        // oplus(x)
    virtual bool terminate() { return false; }
};

static int lm_optimize(LMProblem &P, int iterations) {
    double lambda = -1, ni = 2;
    int nBad = 0, done = 0;
    const double goodUpper = 2. / 3., goodLower = 1. / 3., tau = 1e-5;
    for (int it = 0; it < iterations && !P.terminate(); it++) {
        double currentChi = P.robust_chi2(), tempChi = currentChi;
        const double iniChi = currentChi;
        P.build();
        if (it == 0) { lambda = tau * P.max_diag(); ni = 2; nBad = 0; }
        double rho = 0;
        int qmax = 0;
        do {
            P.push();
            const bool ok2 = P.solve(lambda);
            P.update();
            tempChi = P.robust_chi2();
            if (!ok2) tempChi = std::numeric_limits<double>::max();
            rho = currentChi - tempChi;
            double sc = P.scale(lambda);
            sc += 1e-3;
            rho /= sc;
            if (rho > 0 && std::isfinite(tempChi)) {
                double alpha = 1. - std::pow((2 * rho - 1), 3);
                alpha = std::min(alpha, goodUpper);
                const double scaleFactor = std::max(goodLower, alpha);
                lambda *= scaleFactor;
                ni = 2;
                currentChi = tempChi;
            } else {
                lambda *= ni;
                ni *= 2;
                P.pop();
            }
            qmax++;
        } while (rho < 0 && qmax < 10 && !P.terminate());
        done++;
        if (qmax == 10 || rho == 0) break;                 // Terminate
        if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
        if (nBad >= 3) break;
    }
    return done;
}

// ---------------------------------------------------------------------------------------------------------------
struct PoseProblem : LMProblem {
    int n; const V3 *Xw; const double *obs, *info; const uint8_t *active; Cam cam; bool robust; double delta, dsqr;
    SE3 T, saved;
    double H[36], b[6], x[6];
    std::vector<double> lastChi2;      // chi2 of each edge as of the last computeActiveErrors() (g2o keeps _error on the edge)
    double edge_chi2(int i, double e[2]) const {
        double u, v;
        project(cam, se3_map(T, Xw[i]), u, v);
        e[0] = obs[2 * i] - u; e[1] = obs[2 * i + 1] - v;
        return e[0] * info[i] * e[0] + e[1] * info[i] * e[1];
    }
    double robust_chi2() override {
        double chi = 0;
        for (int i = 0; i < n; i++) {
            if (!active[i]) continue;
            double e[2], c = edge_chi2(i, e);
            lastChi2[i] = c;
            if (robust) { double r[3]; huber(c, delta, dsqr, r); chi += r[0]; } else chi += c;
        }
        return chi;
    }
    void build() override {
        std::memset(H, 0, sizeof H); std::memset(b, 0, sizeof b);
        for (int i = 0; i < n; i++) {
            if (!active[i]) continue;
            double e[2], c = edge_chi2(i, e), r[3] = {c, 1, 0};
            if (robust) huber(c, delta, dsqr, r);
            double J[2][6];
            jac_pose(cam, se3_map(T, Xw[i]), J);
            const double w = info[i];
            for (int a = 0; a < 6; a++) {
                b[a] -= r[1] * (J[0][a] * w * e[0] + J[1][a] * w * e[1]);
                for (int c2 = 0; c2 < 6; c2++) H[a * 6 + c2] += (r[1] * w) * (J[0][a] * J[0][c2] + J[1][a] * J[1][c2]);
            }
        }
    }
    double max_diag() override { double m = 0; for (int j = 0; j < 6; j++) m = std::max(std::fabs(H[j * 7]), m); return m; }
    bool solve(double lambda) override {
        std::vector<double> A(H, H + 36);
        for (int j = 0; j < 6; j++) A[j * 7] += lambda;
        return chol_solve(A, 6, b, x);
    }
    double scale(double lambda) override { double s = 0; for (int j = 0; j < 6; j++) s += x[j] * (lambda * x[j] + b[j]); return s; }
    void push() override { saved = T; }
    void pop() override { T = saved; }
    void update() override { T = se3_mul(se3_exp(x), T); }
};

// ---------------------------------------------------------------------------------------------------------------
struct BAProblem : LMProblem {
    int nKF, nMP, nE;
    std::vector<SE3> T, Tsaved; std::vector<V3> X, Xsaved;
    const uint8_t *fixedKF; const int32_t *eMP, *eKF; const double *obs, *info; Cam cam; double delta, dsqr;
    const volatile uint8_t *stop;
    std::vector<int> poseCol;                     // column block of each non-fixed KF, -1 if fixed
    int nOpt = 0;
    std::vector<double> Hpp, bp, Hll, bl, Hpl, x, Dinv;  // Hpp: nOpt x 36, Hll: nMP x 9, Hpl: nE x 18 (6x3), x: 6nOpt + 3nMP
    std::vector<std::vector<int>> edgesOfPoint;
    std::vector<double> lastChi2;                 // per edge, as of the last computeActiveErrors()
    std::vector<uint8_t> level1;                  // edge moved to level 1 (merge BA second pass): not an active edge any more
    bool robust = true;                           // Huber kernel attached to the edges

    void init() {
        level1.assign(nE, 0);
        poseCol.assign(nKF, -1);
        for (int k = 0; k < nKF; k++) if (!fixedKF[k]) poseCol[k] = nOpt++;
        edgesOfPoint.assign(nMP, {});
        for (int e = 0; e < nE; e++) edgesOfPoint[eMP[e]].push_back(e);
        Hpp.assign((size_t)nOpt * 36, 0); bp.assign((size_t)nOpt * 6, 0); Hll.assign((size_t)nMP * 9, 0); bl.assign((size_t)nMP * 3, 0);
        Hpl.assign((size_t)nE * 18, 0); x.assign((size_t)nOpt * 6 + (size_t)nMP * 3, 0); Dinv.assign((size_t)nMP * 9, 0);
    }
    double edge_err(int e, double r[2], V3 *pc = nullptr) const {
        const V3 p = se3_map(T[eKF[e]], X[eMP[e]]);
        if (pc) *pc = p;
        double u, v;
        project(cam, p, u, v);
        r[0] = obs[2 * e] - u; r[1] = obs[2 * e + 1] - v;
        return r[0] * info[e] * r[0] + r[1] * info[e] * r[1];
    }
    double robust_chi2() override {
        double chi = 0;
        lastChi2.resize(nE);
        for (int e = 0; e < nE; e++) {
            if (level1[e]) continue;                                      // computeActiveErrors() visits active edges only
            double r[2], rho[3];
            lastChi2[e] = edge_err(e, r);
            if (robust) { huber(lastChi2[e], delta, dsqr, rho); chi += rho[0]; }
            else chi += lastChi2[e];
        }
        return chi;
    }
    void build() override {
        std::fill(Hpp.begin(), Hpp.end(), 0.); std::fill(bp.begin(), bp.end(), 0.); std::fill(Hll.begin(), Hll.end(), 0.);
        std::fill(bl.begin(), bl.end(), 0.); std::fill(Hpl.begin(), Hpl.end(), 0.);
        for (int e = 0; e < nE; e++) {
            double r[2], rho[3];
            V3 pc;
            if (level1[e]) continue;
            const double c = edge_err(e, r, &pc);
            if (robust) huber(c, delta, dsqr, rho); else { rho[0] = c; rho[1] = 1; rho[2] = 0; }
            const double w = rho[1] * info[e];
            double B[2][6], A[2][3], R[3][3];
            jac_pose(cam, pc, B);                                        // d e / d pose
            quat_to_R(T[eKF[e]].r, R);
            const double j00 = cam.fx / pc.z, j02 = -cam.fx * pc.x / (pc.z * pc.z), j11 = cam.fy / pc.z, j12 = -cam.fy * pc.y / (pc.z * pc.z);
            for (int k = 0; k < 3; k++) { A[0][k] = -(j00 * R[0][k] + j02 * R[2][k]); A[1][k] = -(j11 * R[1][k] + j12 * R[2][k]); }   // -projectJac * R
            const int p = eMP[e], col = poseCol[eKF[e]];
            for (int a = 0; a < 3; a++) {
                bl[p * 3 + a] -= rho[1] * info[e] * (A[0][a] * r[0] + A[1][a] * r[1]);
                for (int c2 = 0; c2 < 3; c2++) Hll[p * 9 + a * 3 + c2] += w * (A[0][a] * A[0][c2] + A[1][a] * A[1][c2]);
            }
            if (col >= 0) {
                for (int a = 0; a < 6; a++) {
                    bp[col * 6 + a] -= rho[1] * info[e] * (B[0][a] * r[0] + B[1][a] * r[1]);
                    for (int c2 = 0; c2 < 6; c2++) Hpp[col * 36 + a * 6 + c2] += w * (B[0][a] * B[0][c2] + B[1][a] * B[1][c2]);
                    for (int c2 = 0; c2 < 3; c2++) Hpl[(size_t)e * 18 + a * 3 + c2] = w * (B[0][a] * A[0][c2] + B[1][a] * A[1][c2]);
                }
            }
        }
    }
    double max_diag() override {
        double m = 0;
        for (int k = 0; k < nOpt; k++) for (int j = 0; j < 6; j++) m = std::max(std::fabs(Hpp[k * 36 + j * 7]), m);
        for (int p = 0; p < nMP; p++) for (int j = 0; j < 3; j++) m = std::max(std::fabs(Hll[p * 9 + j * 4]), m);
        return m;
    }
    bool solve(double lambda) override {
        const int n = 6 * nOpt;
        std::vector<double> S((size_t)n * n, 0.), bs(bp);
        for (int k = 0; k < nOpt; k++)
            for (int a = 0; a < 6; a++) for (int c = 0; c < 6; c++) S[(size_t)(k * 6 + a) * n + k * 6 + c] = Hpp[k * 36 + a * 6 + c] + (a == c ? lambda : 0.);
        for (int p = 0; p < nMP; p++) {
            double D[9];
            for (int i = 0; i < 9; i++) D[i] = Hll[p * 9 + i];
            D[0] += lambda; D[4] += lambda; D[8] += lambda;
            // 3x3 inverse by cofactors (Eigen Matrix3d::inverse)
            const double c00 = D[4] * D[8] - D[5] * D[7], c01 = D[5] * D[6] - D[3] * D[8], c02 = D[3] * D[7] - D[4] * D[6];
            const double det = D[0] * c00 + D[1] * c01 + D[2] * c02, id = 1.0 / det;
            double *I = &Dinv[p * 9];
            I[0] = c00 * id; I[1] = (D[2] * D[7] - D[1] * D[8]) * id; I[2] = (D[1] * D[5] - D[2] * D[4]) * id;
            I[3] = c01 * id; I[4] = (D[0] * D[8] - D[2] * D[6]) * id; I[5] = (D[2] * D[3] - D[0] * D[5]) * id;
            I[6] = c02 * id; I[7] = (D[1] * D[6] - D[0] * D[7]) * id; I[8] = (D[0] * D[4] - D[1] * D[3]) * id;
            double db[3];
            for (int a = 0; a < 3; a++) db[a] = I[a * 3] * bl[p * 3] + I[a * 3 + 1] * bl[p * 3 + 1] + I[a * 3 + 2] * bl[p * 3 + 2];
            for (int e1 : edgesOfPoint[p]) {
                const int c1 = poseCol[eKF[e1]];
                if (c1 < 0) continue;
                const double *B1 = &Hpl[(size_t)e1 * 18];
                double BD[18];
                for (int a = 0; a < 6; a++) for (int c = 0; c < 3; c++) BD[a * 3 + c] = B1[a * 3] * I[c] + B1[a * 3 + 1] * I[3 + c] + B1[a * 3 + 2] * I[6 + c];
                for (int a = 0; a < 6; a++) bs[c1 * 6 + a] -= B1[a * 3] * db[0] + B1[a * 3 + 1] * db[1] + B1[a * 3 + 2] * db[2];
                for (int e2 : edgesOfPoint[p]) {
                    const int c2 = poseCol[eKF[e2]];
                    if (c2 < 0) continue;
                    const double *B2 = &Hpl[(size_t)e2 * 18];
                    for (int a = 0; a < 6; a++) for (int c = 0; c < 6; c++)
                        S[(size_t)(c1 * 6 + a) * n + c2 * 6 + c] -= BD[a * 3] * B2[c * 3] + BD[a * 3 + 1] * B2[c * 3 + 1] + BD[a * 3 + 2] * B2[c * 3 + 2];
                }
            }
        }
        std::fill(x.begin(), x.end(), 0.);
        if (n > 0 && !chol_solve(S, n, bs.data(), x.data())) return false;
        for (int p = 0; p < nMP; p++) {                                   // xl = Dinv (bl - Hpl^T xp)
            double cl[3] = {bl[p * 3], bl[p * 3 + 1], bl[p * 3 + 2]};
            for (int e : edgesOfPoint[p]) {
                const int c1 = poseCol[eKF[e]];
                if (c1 < 0) continue;
                const double *B1 = &Hpl[(size_t)e * 18];
                for (int c = 0; c < 3; c++) for (int a = 0; a < 6; a++) cl[c] -= B1[a * 3 + c] * x[c1 * 6 + a];
            }
            const double *I = &Dinv[p * 9];
            for (int a = 0; a < 3; a++) x[n + p * 3 + a] = I[a * 3] * cl[0] + I[a * 3 + 1] * cl[1] + I[a * 3 + 2] * cl[2];
        }
        return true;
    }
    double scale(double lambda) override {
        double s = 0;
        const int n = 6 * nOpt;
        for (int j = 0; j < n; j++) s += x[j] * (lambda * x[j] + bp[j]);
        for (int j = 0; j < 3 * nMP; j++) s += x[n + j] * (lambda * x[n + j] + bl[j]);
        return s;
    }
    void push() override { Tsaved = T; Xsaved = X; }
    void pop() override { T = Tsaved; X = Xsaved; }
    void update() override {
        const int n = 6 * nOpt;
        for (int k = 0; k < nKF; k++) if (poseCol[k] >= 0) T[k] = se3_mul(se3_exp(&x[poseCol[k] * 6]), T[k]);
        for (int p = 0; p < nMP; p++) { X[p].x += x[n + p * 3]; X[p].y += x[n + p * 3 + 1]; X[p].z += x[n + p * 3 + 2]; }
    }
    bool terminate() override { return stop && *stop; }
};

}  // namespace orcopt

using namespace orcopt;


// ---------------------------------------------------------------------------------------------------------------
// g2o::Sim3 (G/types/sim3.h) and the one-vertex Sim3 problems Optimizer::OptimizeSim3 (R/lib_src/Optimizer.cc:1920-2167) and
// Optimizer::OptimizeCloudSim3 (:2169-2471) build: all points are fixed vertices, the edges have no analytic linearizeOplus
// (commented out in OptimizableTypes.h:186,205,224,249,267,292), so g2o differentiates numerically
// (G/core/base_binary_edge.hpp:131-203: central differences, delta 1e-9, through VertexSim3Expmap::oplusImpl).
// ---------------------------------------------------------------------------------------------------------------
struct Sim3 { Quat r; V3 t; double s; };
static inline V3 sim3_map(const Sim3 &S, V3 p) { const V3 r = rotate(S.r, p); return {S.s * r.x + S.t.x, S.s * r.y + S.t.y, S.s * r.z + S.t.z}; }   // sim3.h:144-146
static inline Sim3 sim3_mul(const Sim3 &a, const Sim3 &b) {                                             // sim3.h:266-272 (no re-normalisation)
    Sim3 o;
    o.r = qmul(a.r, b.r);
    const V3 rt = rotate(a.r, b.t);
    o.t = {a.s * rt.x + a.t.x, a.s * rt.y + a.t.y, a.s * rt.z + a.t.z};
    o.s = a.s * b.s;
    return o;
}
static inline Sim3 sim3_inverse(const Sim3 &a) {                                                         // sim3.h:233-236
    const Quat c{-a.r.x, -a.r.y, -a.r.z, a.r.w};
    const double k = -1. / a.s;
    return Sim3{c, rotate(c, V3{k * a.t.x, k * a.t.y, k * a.t.z}), 1. / a.s};
}
static Sim3 sim3_exp(const double u[7]) {                                                                // sim3.h:70-142
    const double wx = u[0], wy = u[1], wz = u[2], sigma = u[6];
    const double theta = std::sqrt(wx * wx + wy * wy + wz * wz);
    const double O[3][3] = {{0, -wz, wy}, {wz, 0, -wx}, {-wy, wx, 0}};
    double O2[3][3], R[3][3], W[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { O2[i][j] = 0; for (int k = 0; k < 3; k++) O2[i][j] += O[i][k] * O[k][j]; }
    const double s = std::exp(sigma), eps = 0.00001;
    double A, B, C;
    auto setR = [&](double a, double b) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i][j] = ((i == j) + a * O[i][j]) + b * O2[i][j]; };
    if (std::fabs(sigma) < eps) {
        C = 1;
        if (theta < eps) { A = 1. / 2.; B = 1. / 6.; setR(1, 1); }
        else {
            const double theta2 = theta * theta;
            A = (1 - std::cos(theta)) / theta2; B = (theta - std::sin(theta)) / (theta2 * theta);
            setR(std::sin(theta) / theta, (1 - std::cos(theta)) / (theta * theta));
        }
    } else {
        C = (s - 1) / sigma;
        if (theta < eps) {
            const double sigma2 = sigma * sigma;
            A = ((sigma - 1) * s + 1) / sigma2; B = ((0.5 * sigma2 - sigma + 1) * s) / (sigma2 * sigma);
            setR(1, 1);
        } else {
            setR(std::sin(theta) / theta, (1 - std::cos(theta)) / (theta * theta));
            const double a = s * std::sin(theta), b = s * std::cos(theta), theta2 = theta * theta, sigma2 = sigma * sigma, c = theta2 + sigma2;
            A = (a * sigma + (1 - b) * theta) / (theta * c);
            B = (C - ((b - 1) * sigma + a * theta) / c) * 1. / theta2;
        }
    }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) W[i][j] = (A * O[i][j] + B * O2[i][j]) + C * (i == j);
    Sim3 S;
    S.r = quat_from_R(R);
    S.t = {W[0][0] * u[3] + W[0][1] * u[4] + W[0][2] * u[5], W[1][0] * u[3] + W[1][1] * u[4] + W[1][2] * u[5],
           W[2][0] * u[3] + W[2][1] * u[4] + W[2][2] * u[5]};
    S.s = s;
    return S;
}

struct Sim3Problem : LMProblem {
    int n = 0;
    const int32_t *pairOf = nullptr; const double *Sc1w = nullptr, *Sc2w = nullptr;   // null: EdgeSim3ProjectXYZ / EdgeInverseSim3ProjectXYZ
    const float *P1c, *P2c, *obs1, *obs2, *w1, *w2;
    std::vector<uint8_t> on12, on21;          // edge present in the active graph
    Cam cam1, cam2; bool fixScale = false, robust = false; double delta = 0, dsqr = 0;
    Sim3 est, saved;
    double H[49], b[7], x[7];
    std::vector<double> chi12, chi21;         // chi2 as of the last computeActiveErrors()

    // computeError of the edge pair of correspondence i under the vertex estimate S
    void errors(int i, const Sim3 &S, bool do12, bool do21, double e12[2], double e21[2]) const {
        const int p = pairOf ? pairOf[i] : 0;
        if (do12) {
            Sim3 T = S;
            if (Sc1w) T = sim3_mul(sim3_mul(sim3_from(Sc1w + 8 * (size_t)p), S), sim3_inverse(sim3_from(Sc2w + 8 * (size_t)p)));   // gSc1w * est * gSc2w.inverse()
            double u, v;
            project(cam1, sim3_map(T, V3{P2c[3 * i], P2c[3 * i + 1], P2c[3 * i + 2]}), u, v);
            e12[0] = (double)obs1[2 * i] - u; e12[1] = (double)obs1[2 * i + 1] - v;
        }
        if (do21) {
            Sim3 T = sim3_inverse(S);
            if (Sc1w) T = sim3_mul(sim3_mul(sim3_from(Sc2w + 8 * (size_t)p), sim3_inverse(S)), sim3_inverse(sim3_from(Sc1w + 8 * (size_t)p)));
            double u, v;
            project(cam2, sim3_map(T, V3{P1c[3 * i], P1c[3 * i + 1], P1c[3 * i + 2]}), u, v);
            e21[0] = (double)obs2[2 * i] - u; e21[1] = (double)obs2[2 * i + 1] - v;
        }
    }
    static Sim3 sim3_from(const double *S) { return Sim3{{S[0], S[1], S[2], S[3]}, {S[4], S[5], S[6]}, S[7]}; }
    Sim3 oplus(const Sim3 &S, const double *upd) const {             // VertexSim3Expmap::oplusImpl (OptimizableTypes.h:156-164)
        double u[7];
        for (int k = 0; k < 7; k++) u[k] = upd[k];
        if (fixScale) u[6] = 0;
        return sim3_mul(sim3_exp(u), S);
    }
    static double chi2_of(const double e[2], double w) { return e[0] * w * e[0] + e[1] * w * e[1]; }
    double robust_chi2() override {
        double chi = 0;
        for (int i = 0; i < n; i++) {
            double e12[2], e21[2];
            errors(i, est, on12[i], on21[i], e12, e21);
            if (on12[i]) { const double c = chi2_of(e12, w1[i]); chi12[i] = c; double r[3] = {c, 1, 0}; if (robust) huber(c, delta, dsqr, r); chi += r[0]; }
            if (on21[i]) { const double c = chi2_of(e21, w2[i]); chi21[i] = c; double r[3] = {c, 1, 0}; if (robust) huber(c, delta, dsqr, r); chi += r[0]; }
        }
        return chi;
    }
    void accumulate(const double e[2], const double J[2][7], double w) {
        double c = chi2_of(e, w), r[3] = {c, 1, 0};
        if (robust) huber(c, delta, dsqr, r);
        for (int a = 0; a < 7; a++) {
            b[a] -= r[1] * (J[0][a] * w * e[0] + J[1][a] * w * e[1]);
            for (int c2 = 0; c2 < 7; c2++) H[a * 7 + c2] += (r[1] * w) * (J[0][a] * J[0][c2] + J[1][a] * J[1][c2]);
        }
    }
    void build() override {
        std::memset(H, 0, sizeof H); std::memset(b, 0, sizeof b);
        const double dlt = 1e-9, scalar = 1.0 / (2 * dlt);
        for (int i = 0; i < n; i++) {
            if (!on12[i] && !on21[i]) continue;
            double e12[2], e21[2], J12[2][7], J21[2][7];
            for (int d = 0; d < 7; d++) {
                double add[7] = {0, 0, 0, 0, 0, 0, 0}, p12[2], p21[2], m12[2], m21[2];
                add[d] = dlt;
                errors(i, oplus(est, add), on12[i], on21[i], p12, p21);
                add[d] = -dlt;
                errors(i, oplus(est, add), on12[i], on21[i], m12, m21);
                for (int r = 0; r < 2; r++) { J12[r][d] = scalar * (p12[r] - m12[r]); J21[r][d] = scalar * (p21[r] - m21[r]); }
            }
            errors(i, est, on12[i], on21[i], e12, e21);
            if (on12[i]) accumulate(e12, J12, w1[i]);
            if (on21[i]) accumulate(e21, J21, w2[i]);
        }
    }
    double max_diag() override { double m = 0; for (int j = 0; j < 7; j++) m = std::max(std::fabs(H[j * 8]), m); return m; }
    bool solve(double lambda) override {
        std::vector<double> A(H, H + 49);
        for (int j = 0; j < 7; j++) A[j * 8] += lambda;
        return chol_solve(A, 7, b, x);
    }
    double scale(double lambda) override { double s = 0; for (int j = 0; j < 7; j++) s += x[j] * (lambda * x[j] + b[j]); return s; }
    void push() override { saved = est; }
    void pop() override { est = saved; }
    void update() override { est = oplus(est, x); }
};

extern "C" {

// Optimizer::PoseOptimization on flat arrays (one entry per feature that holds a map point, in feature order).
// Xw: n x 3 float (GetWorldPos), obs: n x 2 float (mvKeysUn pt), invSigma2: n float, K4: fx fy cx cy, Tcw7 in/out
// (qx qy qz qw tx ty tz), outlier: n u8 out.  Returns nInitialCorrespondences - nBad (0 and no change if n < 3).
int orc_pose_optimization(const float *Xw, const float *obs, const float *invSigma2, int n, const float *K4, float *Tcw7,
                          uint8_t *outlier) {
    for (int i = 0; i < n; i++) outlier[i] = 0;
    if (n < 3) return 0;
    std::vector<V3> X(n);
    std::vector<double> o(2 * (size_t)n), w(n);
    for (int i = 0; i < n; i++) {
        X[i] = {Xw[3 * i], Xw[3 * i + 1], Xw[3 * i + 2]};
        o[2 * i] = obs[2 * i]; o[2 * i + 1] = obs[2 * i + 1]; w[i] = invSigma2[i];
    }
    std::vector<uint8_t> active(n, 1);
    PoseProblem P;
    P.lastChi2.assign(n, 0.);
    P.n = n; P.Xw = X.data(); P.obs = o.data(); P.info = w.data(); P.active = active.data();
    P.cam = {K4[0], K4[1], K4[2], K4[3]};
    const float deltaMono = (float)std::sqrt(5.991);
    P.delta = deltaMono; P.dsqr = P.delta * P.delta; P.robust = true;
    const SE3 T0 = se3_from_float7(Tcw7);
    const float chi2Mono[4] = {5.991f, 5.991f, 5.991f, 5.991f};
    int nBad = 0;
    for (int it = 0; it < 4; it++) {
        P.T = T0;                                                   // the estimate is reset to pFrame->GetPose() every round
        int nActive = 0;
        for (int i = 0; i < n; i++) nActive += active[i];
        if (nActive > 0) lm_optimize(P, 10);                        // initializeOptimization(0) + optimize(10)
        nBad = 0;
        for (int i = 0; i < n; i++) {
            // Optimizer.cc:917-926: edges that were outliers (inactive) get computeError() at the final pose; active edges
            // keep the error of the LAST computeActiveErrors(), which after 10 rejected trials is the rejected state.
            double e[2];
            const float chi2 = (float)(outlier[i] ? P.edge_chi2(i, e) : P.lastChi2[i]);
            if (chi2 > chi2Mono[it]) { outlier[i] = 1; active[i] = 0; nBad++; }
            else { outlier[i] = 0; active[i] = 1; }
        }
        if (it == 2) P.robust = false;
        if (n < 10) break;
    }
    se3_to_float7(P.T, Tcw7);
    return n - nBad;
}

// Optimizer::LocalBundleAdjustment on a flattened graph.  kfPose: nKF x 7 float in/out (only non-fixed are written),
// kfFixed: nKF u8, mpPos: nMP x 3 float in/out, edges: (mp, kf, obs x/y float, invSigma2 float), K4, stop flag (may be null).
// eraseOut: nE u8 (chi2 > 5.991 || depth <= 0 after optimisation).  Returns the number of LM iterations run, -1 if aborted
// before optimising (stop flag set, Optimizer.cc:1274-1276).
int orc_local_ba(int nKF, float *kfPose, const uint8_t *kfFixed, int nMP, float *mpPos, int nE, const int32_t *eMP,
                 const int32_t *eKF, const float *eObs, const float *eInvSigma2, const float *K4, const volatile uint8_t *stop,
                 uint8_t *eraseOut) {
    if (stop && *stop) return -1;
    BAProblem P;
    P.nKF = nKF; P.nMP = nMP; P.nE = nE; P.fixedKF = kfFixed; P.eMP = eMP; P.eKF = eKF; P.stop = stop;
    P.T.resize(nKF); P.X.resize(nMP);
    for (int k = 0; k < nKF; k++) P.T[k] = se3_from_float7(kfPose + 7 * k);
    for (int p = 0; p < nMP; p++) P.X[p] = {mpPos[3 * p], mpPos[3 * p + 1], mpPos[3 * p + 2]};
    std::vector<double> o(2 * (size_t)nE), w(nE);
    for (int e = 0; e < nE; e++) { o[2 * e] = eObs[2 * e]; o[2 * e + 1] = eObs[2 * e + 1]; w[e] = eInvSigma2[e]; }
    P.obs = o.data(); P.info = w.data();
    P.cam = {K4[0], K4[1], K4[2], K4[3]};
    const float thHuberMono = (float)std::sqrt(5.991);
    P.delta = thHuberMono; P.dsqr = P.delta * P.delta;
    P.init();
    const int its = lm_optimize(P, 10);
    for (int e = 0; e < nE; e++) {
        double r[2];
        V3 pc;
        P.edge_err(e, r, &pc);                                            // isDepthPositive() uses the current estimates
        const double chi2 = its > 0 ? P.lastChi2[e] : P.edge_err(e, r);   // e->chi2() = error of the last computeActiveErrors()
        eraseOut[e] = (chi2 > 5.991 || !(pc.z > 0.0)) ? 1 : 0;
    }
    for (int k = 0; k < nKF; k++) if (!kfFixed[k]) se3_to_float7(P.T[k], kfPose + 7 * k);
    for (int p = 0; p < nMP; p++) { mpPos[3 * p] = (float)P.X[p].x; mpPos[3 * p + 1] = (float)P.X[p].y; mpPos[3 * p + 2] = (float)P.X[p].z; }
    return its;
}

// Optimizer::LocalBundleAdjustment(KeyFrame *pMainKF, vpAdjustKF, vpFixedKF, bool *pbStopFlag), monocular edges
// (R/lib_src/Optimizer.cc:3768-4183), on the same flattened graph as orc_local_ba.  its2[0] / its2[1] receive the LM iterations
// of the two optimize() calls.  Returns -1 when the stop flag is already set (:3982-3984).
int orc_merge_ba(int nKF, float *kfPose, const uint8_t *kfFixed, int nMP, float *mpPos, int nE, const int32_t *eMP, const int32_t *eKF,
                 const float *eObs, const float *eInvSigma2, const float *K4, const volatile uint8_t *stop, uint8_t *eraseOut, int32_t *its2) {
    if (its2) its2[0] = its2[1] = 0;
    if (stop && *stop) return -1;
    BAProblem P;
    P.nKF = nKF; P.nMP = nMP; P.nE = nE; P.fixedKF = kfFixed; P.eMP = eMP; P.eKF = eKF; P.stop = stop;
    P.T.resize(nKF); P.X.resize(nMP);
    for (int k = 0; k < nKF; k++) P.T[k] = se3_from_float7(kfPose + 7 * k);
    for (int p = 0; p < nMP; p++) P.X[p] = {mpPos[3 * p], mpPos[3 * p + 1], mpPos[3 * p + 2]};
    std::vector<double> o(2 * (size_t)nE), w(nE);
    for (int e = 0; e < nE; e++) { o[2 * e] = eObs[2 * e]; o[2 * e + 1] = eObs[2 * e + 1]; w[e] = eInvSigma2[e]; }
    P.obs = o.data(); P.info = w.data();
    P.cam = {K4[0], K4[1], K4[2], K4[3]};
    const float thHuber2D = (float)std::sqrt(5.99);                       // :3871
    P.delta = thHuber2D; P.dsqr = P.delta * P.delta;
    P.init();
    const int its1 = lm_optimize(P, 5);                                   // :3986-3987
    int its = its1, itsSecond = 0;
    const bool bDoMore = !(stop && *stop);
    if (bDoMore) {                                                        // :3996-4031
        for (int e = 0; e < nE; e++) {
            double r[2];
            V3 pc;
            P.edge_err(e, r, &pc);
            const double chi2 = its1 > 0 ? P.lastChi2[e] : P.edge_err(e, r);
            if (chi2 > 5.991 || !(pc.z > 0.0)) P.level1[e] = 1;           // e->setLevel(1)
        }
        P.robust = false;                                                 // e->setRobustKernel(0)
        itsSecond = lm_optimize(P, 10);                                   // initializeOptimization(0); optimize(10)
        its += itsSecond;
    }
    for (int e = 0; e < nE; e++) {                                        // :4042-4056
        double r[2];
        V3 pc;
        P.edge_err(e, r, &pc);
        const double chi2 = its > 0 ? P.lastChi2[e] : P.edge_err(e, r);
        eraseOut[e] = (chi2 > 5.991 || !(pc.z > 0.0)) ? 1 : 0;
    }
    for (int k = 0; k < nKF; k++) if (!kfFixed[k]) se3_to_float7(P.T[k], kfPose + 7 * k);
    for (int p = 0; p < nMP; p++) { mpPos[3 * p] = (float)P.X[p].x; mpPos[3 * p + 1] = (float)P.X[p].y; mpPos[3 * p + 2] = (float)P.X[p].z; }
    if (its2) { its2[0] = its1; its2[1] = itsSecond; }
    return its;
}


// Optimizer::BundleAdjustment (R/lib_src/Optimizer.cc:54-351), monocular edges, on the flattened graph of orc_local_ba: one
// optimize(nIterations), Huber(sqrt(5.99)) on every edge when bRobust.  Returns the LM iterations run.
int orc_bundle_adjustment(int nKF, float *kfPose, const uint8_t *kfFixed, int nMP, float *mpPos, int nE, const int32_t *eMP, const int32_t *eKF,
                          const float *eObs, const float *eInvSigma2, const float *K4, const volatile uint8_t *stop, int nIterations, int bRobust) {
    BAProblem P;
    P.nKF = nKF; P.nMP = nMP; P.nE = nE; P.fixedKF = kfFixed; P.eMP = eMP; P.eKF = eKF; P.stop = stop;
    P.T.resize(nKF); P.X.resize(nMP);
    for (int k = 0; k < nKF; k++) P.T[k] = se3_from_float7(kfPose + 7 * k);
    for (int p = 0; p < nMP; p++) P.X[p] = {mpPos[3 * p], mpPos[3 * p + 1], mpPos[3 * p + 2]};
    std::vector<double> o(2 * (size_t)nE), w(nE);
    for (int e = 0; e < nE; e++) { o[2 * e] = eObs[2 * e]; o[2 * e + 1] = eObs[2 * e + 1]; w[e] = eInvSigma2[e]; }
    P.obs = o.data(); P.info = w.data();
    P.cam = {K4[0], K4[1], K4[2], K4[3]};
    const float thHuber2D = (float)std::sqrt(5.99);                       // :122
    P.delta = thHuber2D; P.dsqr = P.delta * P.delta;
    P.init();
    P.robust = bRobust != 0;
    const int its = lm_optimize(P, nIterations);                          // :258-260
    for (int k = 0; k < nKF; k++) if (!kfFixed[k]) se3_to_float7(P.T[k], kfPose + 7 * k);
    for (int p = 0; p < nMP; p++) { mpPos[3 * p] = (float)P.X[p].x; mpPos[3 * p + 1] = (float)P.X[p].y; mpPos[3 * p + 2] = (float)P.X[p].z; }
    return its;
}

// Sim3Solver::ComputeInliersNum (R/lib_src/Sim3Solver.cc:564-664) on flat arrays (layout: include/rumi_opt.h, rumi_sim3_inliers).
float orc_sim3_inliers(int nPairs, const int32_t *pairStart, const int32_t *pairDenom, const double *Sc1w2, const double *Sc2w1, const float *K1,
                       const float *K2, const float *X1, const float *X2, const float *kp1, const float *kp2, const float *sigma1,
                       const float *sigma2, const uint8_t *edge1, const uint8_t *edge2, uint8_t *inlier, float *ratioOut) {
    if (nPairs == 0) return 0.f;
    std::vector<float> ratios;
    auto reproj2 = [](const double *S, const float *K, const float *X, const float *kp) {
        const Quat q{S[0], S[1], S[2], S[3]};
        const V3 r = rotate(q, V3{(double)X[0], (double)X[1], (double)X[2]});                 // g2o::Sim3::map: s*(r*xyz) + t
        const double s = S[7];
        const double px = s * r.x + S[4], py = s * r.y + S[5], pz = s * r.z + S[6];
        const float u = (float)(K[0] * px / pz + K[2]), v = (float)(K[1] * py / pz + K[3]);   // Pinhole::project(Vector3d).cast<float>()
        const float dx = kp[0] - u, dy = kp[1] - v;
        return dx * dx + dy * dy;                                                             // dist.dot(dist)
    };
    for (int p = 0; p < nPairs; p++) {
        int nIn = 0;
        for (int i = pairStart[p]; i < pairStart[p + 1]; i++) {
            const float err1 = reproj2(Sc1w2 + (size_t)p * 8, K1, X2 + (size_t)i * 3, kp1 + (size_t)i * 2);
            const float err2 = reproj2(Sc2w1 + (size_t)p * 8, K2, X1 + (size_t)i * 3, kp2 + (size_t)i * 2);
            const bool in = (err1 < (2 * 9.210 * sigma1[i]) || edge2[i]) && (err2 < (2 * 9.210 * sigma2[i]) || edge1[i]);
            inlier[i] = in;
            nIn += in;
        }
        ratios.push_back(pairDenom[p] ? (float)nIn / (float)pairDenom[p] : 0.f);
    }
    if (ratioOut) std::memcpy(ratioOut, ratios.data(), ratios.size() * sizeof(float));
    std::sort(ratios.begin(), ratios.end());
    return ratios[ratios.size() / 2];
}

// ---- Sim3Solver::iterate, one hypothesis at a time (R/lib_src/Sim3Solver.cc:159-404; layout: include/rumi_opt.h, rumi_sim3_ransac) ----------
// The minimal sets: nHyp x 3 indices drawn as :176-191 does, with DUtils::Random::RandomInt (Thirdparty/DBoW2/DUtils/Random.cpp:47-50) on this
// process's glibc rand() (the caller seeds it, as DUtils::Random::SeedRandOnce(0) does upstream).
void orc_sim3_draw_triples(int n, int nHyp, int32_t *triples) {
    std::vector<int32_t> all(n), avail;
    for (int i = 0; i < n; i++) all[i] = i;                                  // mvAllIndices
    for (int h = 0; h < nHyp; h++) {
        avail = all;
        for (int i = 0; i < 3; i++) {
            const int d = (int)avail.size();                                 // RandomInt(0, size - 1)
            const int randi = int(((double)rand() / ((double)RAND_MAX + 1.0)) * d);
            triples[h * 3 + i] = avail[randi];
            avail[randi] = avail.back();
            avail.pop_back();
        }
    }
}

// dominant eigenvector of a symmetric 4x4 matrix (Horn 1987 step 4); classical Jacobi: annihilate the largest off-diagonal entry until none is left.
// Upstream: Eigen::EigenSolver<Matrix4f> (:477-481) — a third-party routine that is not in the tree; its result is the same vector up to sign and float
// rounding, the rotation built from it does not depend on the sign.
static void dominant_eigvec4(double a[4][4], double out[4]) {
    double v[4][4] = {};
    for (int i = 0; i < 4; i++) v[i][i] = 1;
    for (int it = 0; it < 200; it++) {
        int p = 0, q = 1;
        double big = 0;
        for (int i = 0; i < 4; i++) for (int j = i + 1; j < 4; j++) if (std::fabs(a[i][j]) > big) { big = std::fabs(a[i][j]); p = i; q = j; }
        if (big < 1e-290) break;
        const double phi = 0.5 * std::atan2(2 * a[p][q], a[q][q] - a[p][p]);
        const double c = std::cos(phi), sn = std::sin(phi);
        for (int k = 0; k < 4; k++) { const double x = a[k][p], y = a[k][q]; a[k][p] = c * x - sn * y; a[k][q] = sn * x + c * y; }
        for (int k = 0; k < 4; k++) { const double x = a[p][k], y = a[q][k]; a[p][k] = c * x - sn * y; a[q][k] = sn * x + c * y; }
        for (int k = 0; k < 4; k++) { const double x = v[k][p], y = v[k][q]; v[k][p] = c * x - sn * y; v[k][q] = sn * x + c * y; }
        if (big < 1e-18 * (std::fabs(a[0][0]) + std::fabs(a[1][1]) + std::fabs(a[2][2]) + std::fabs(a[3][3]))) break;
    }
    int best = 0;
    for (int i = 1; i < 4; i++) if (a[i][i] > a[best][best]) best = i;
    for (int k = 0; k < 4; k++) out[k] = v[k][best];
}

void orc_sim3_ransac(int n, const float *X1, const float *X2, const float *sigma2_1, const float *sigma2_2, const float *K1, const float *K2, int fixScale,
                     int nHyp, const int32_t *triples,
                     int nPairs, const int32_t *pairStart, const int32_t *pairDenom, const double *Sc1w1, const double *Sc2w2, const double *Skf1w,
                     const double *Skf2w, const float *sK1, const float *sK2, const float *sX1, const float *sX2, const float *kp1, const float *kp2,
                     const float *sg1, const float *sg2, const uint8_t *e1, const uint8_t *e2,
                     float *T12out, int32_t *nInOut, uint8_t *inlierOut, float *ratioOut, float *medianOut) {
    for (int h = 0; h < nHyp; h++) {
        // ComputeSim3 :437-540 (float members, double where upstream declares doubles)
        float P1[3][3], P2[3][3], O1[3], O2[3], Pr1[3][3], Pr2[3][3];
        for (int i = 0; i < 3; i++) for (int r = 0; r < 3; r++) { P1[r][i] = X1[triples[h * 3 + i] * 3 + r]; P2[r][i] = X2[triples[h * 3 + i] * 3 + r]; }
        for (int r = 0; r < 3; r++) {
            O1[r] = (P1[r][0] + P1[r][1] + P1[r][2]) / 3.f; O2[r] = (P2[r][0] + P2[r][1] + P2[r][2]) / 3.f;
            for (int i = 0; i < 3; i++) { Pr1[r][i] = P1[r][i] - O1[r]; Pr2[r][i] = P2[r][i] - O2[r]; }
        }
        float M[3][3];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { float acc = 0; for (int k = 0; k < 3; k++) acc += Pr2[r][k] * Pr1[c][k]; M[r][c] = acc; }
        const double N11 = M[0][0] + M[1][1] + M[2][2], N12 = M[1][2] - M[2][1], N13 = M[2][0] - M[0][2], N14 = M[0][1] - M[1][0];
        const double N22 = M[0][0] - M[1][1] - M[2][2], N23 = M[0][1] + M[1][0], N24 = M[2][0] + M[0][2];
        const double N33 = -M[0][0] + M[1][1] - M[2][2], N34 = M[1][2] + M[2][1], N44 = -M[0][0] - M[1][1] + M[2][2];
        const float Nf[4][4] = {{(float)N11, (float)N12, (float)N13, (float)N14}, {(float)N12, (float)N22, (float)N23, (float)N24},
                                {(float)N13, (float)N23, (float)N33, (float)N34}, {(float)N14, (float)N24, (float)N34, (float)N44}};
        double Nd[4][4], ev[4];
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) Nd[i][j] = Nf[i][j];
        dominant_eigvec4(Nd, ev);
        float vec[3] = {(float)ev[1], (float)ev[2], (float)ev[3]};
        const float e0 = (float)ev[0];
        const float nrm = std::sqrt(vec[0] * vec[0] + vec[1] * vec[1] + vec[2] * vec[2]);
        const bool valid = !(vec[0] == 0 && vec[1] == 0 && vec[2] == 0);
        float R[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, s12 = 1.f, t12[3] = {0, 0, 0};
        if (valid) {
            const double ang = std::atan2((double)nrm, (double)e0);
            const float f = (float)(2 * ang);
            for (int k = 0; k < 3; k++) vec[k] = vec[k] * f / nrm;
            const float thSq = vec[0] * vec[0] + vec[1] * vec[1] + vec[2] * vec[2];           // Sophus::SO3f::exp
            float im, re;
            if (thSq < 1e-5f * 1e-5f) { const float th4 = thSq * thSq; im = 0.5f - (1.f / 48.f) * thSq + (1.f / 3840.f) * th4; re = 1.f - (1.f / 8.f) * thSq + (1.f / 384.f) * th4; }
            else { const float th = std::sqrt(thSq); im = std::sin(0.5f * th) / th; re = std::cos(0.5f * th); }
            const float q[4] = {im * vec[0], im * vec[1], im * vec[2], re};                 // x y z w
            const float tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
            R[0][0] = 1 - (ty * q[1] + tz * q[2]); R[0][1] = ty * q[0] - tz * q[3]; R[0][2] = tz * q[0] + ty * q[3];
            R[1][0] = ty * q[0] + tz * q[3]; R[1][1] = 1 - (tx * q[0] + tz * q[2]); R[1][2] = tz * q[1] - tx * q[3];
            R[2][0] = tz * q[0] - ty * q[3]; R[2][1] = tz * q[1] + tx * q[3]; R[2][2] = 1 - (tx * q[0] + ty * q[1]);
            if (!fixScale) {
                float nom = 0, den = 0;
                for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) {
                    float p3 = 0;
                    for (int k = 0; k < 3; k++) p3 += R[r][k] * Pr2[k][c];
                    nom += Pr1[r][c] * p3; den += p3 * p3;
                }
                s12 = (float)((double)nom / (double)den);
            }
            for (int r = 0; r < 3; r++) { float acc = 0; for (int k = 0; k < 3; k++) acc += (s12 * R[r][k]) * O2[k]; t12[r] = O1[r] - acc; }
        }
        float T12[3][4], T21[3][4];
        const float sinv = (float)(1.0 / s12);
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) { T12[r][c] = s12 * R[r][c]; T21[r][c] = sinv * R[c][r]; } T12[r][3] = t12[r]; }
        for (int r = 0; r < 3; r++) { float acc = 0; for (int k = 0; k < 3; k++) acc += T21[r][k] * t12[k]; T21[r][3] = -acc; }
        float *To = T12out + (size_t)h * 16;
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) To[r * 3 + c] = R[r][c];
        To[9] = t12[0]; To[10] = t12[1]; To[11] = t12[2]; To[12] = s12; To[13] = valid ? 1.f : 0.f; To[14] = To[15] = 0.f;
        // CheckInliers :542-562
        int nIn = 0;
        for (int i = 0; i < n; i++) {
            const float *a = X1 + (size_t)i * 3, *b = X2 + (size_t)i * 3;
            auto proj = [](const float *K, const float *p, float &u, float &v) { u = K[0] * p[0] / p[2] + K[2]; v = K[1] * p[1] / p[2] + K[3]; };   // Pinhole.cpp:43-49
            float u1, v1, u2, v2, pu, pv, ru, rv, p[3], r[3];
            proj(K1, a, u1, v1); proj(K2, b, u2, v2);
            for (int k = 0; k < 3; k++) {
                p[k] = T12[k][0] * b[0] + T12[k][1] * b[1] + T12[k][2] * b[2] + T12[k][3];
                r[k] = T21[k][0] * a[0] + T21[k][1] * a[1] + T21[k][2] * a[2] + T21[k][3];
            }
            proj(K1, p, pu, pv); proj(K2, r, ru, rv);
            const float err1 = (u1 - pu) * (u1 - pu) + (v1 - pv) * (v1 - pv), err2 = (ru - u2) * (ru - u2) + (rv - v2) * (rv - v2);
            const size_t m1 = (size_t)(9.210 * sigma2_1[i]), m2 = (size_t)(9.210 * sigma2_2[i]);      // Sim3Solver.h:77-78 vector<size_t>
            const bool in = err1 < m1 && err2 < m2;
            if (inlierOut) inlierOut[(size_t)h * n + i] = in;
            nIn += in;
        }
        nInOut[h] = nIn;
        if (nPairs > 0) {                                                                          // :338-348
            double Rd[3][3];
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Rd[r][c] = R[r][c];
            const Sim3 Sc1c2{quat_from_R(Rd), {(double)t12[0], (double)t12[1], (double)t12[2]}, (double)s12};
            auto from8 = [](const double *S) { return Sim3{{S[0], S[1], S[2], S[3]}, {S[4], S[5], S[6]}, S[7]}; };
            const Sim3 Sw1w2 = sim3_mul(sim3_mul(sim3_inverse(from8(Skf1w)), Sc1c2), from8(Skf2w));
            std::vector<double> A((size_t)nPairs * 8), B((size_t)nPairs * 8);
            auto to8 = [](const Sim3 &S, double *o) { o[0] = S.r.x; o[1] = S.r.y; o[2] = S.r.z; o[3] = S.r.w; o[4] = S.t.x; o[5] = S.t.y; o[6] = S.t.z; o[7] = S.s; };
            for (int p = 0; p < nPairs; p++) {
                to8(sim3_mul(from8(Sc1w1 + (size_t)p * 8), Sw1w2), &A[(size_t)p * 8]);              // gSc1w2 :621
                to8(sim3_mul(from8(Sc2w2 + (size_t)p * 8), sim3_inverse(Sw1w2)), &B[(size_t)p * 8]); // gSc2w1 :620
            }
            std::vector<uint8_t> inl((size_t)std::max(pairStart[nPairs], 1));
            medianOut[h] = orc_sim3_inliers(nPairs, pairStart, pairDenom, A.data(), B.data(), sK1, sK2, sX1, sX2, kp1, kp2, sg1, sg2, e1, e2, inl.data(),
                                            ratioOut ? ratioOut + (size_t)h * nPairs : nullptr);
        }
    }
}

// Optimizer::OptimizeSim3 / OptimizeCloudSim3 on flat arrays (layout: include/rumi_opt.h, rumi_optimize_sim3).  One entry per
// correspondence that reaches "nCorrespondences++" (:2035 / :2316).  status: 0 inlier, 1 removed after the first optimize(5),
// 2 fails the final test, 3 kept but uncounted (one of its two edges absent, :2450-2451).  res = {nIn, nBad, early return}.
int orc_optimize_sim3(int n, const int32_t *pairOf, const double *Sc1w, const double *Sc2w, const float *P1c, const float *P2c,
                      const float *obs1, const float *obs2, const float *w1, const float *w2, const uint8_t *skip12, const uint8_t *skip21,
                      const float *K1, const float *K2, float th2, int fixScale, int robustFirst, double *S8, uint8_t *status, int32_t *res) {
    Sim3Problem P;
    P.n = n; P.pairOf = pairOf; P.Sc1w = Sc1w; P.Sc2w = Sc2w; P.P1c = P1c; P.P2c = P2c; P.obs1 = obs1; P.obs2 = obs2; P.w1 = w1; P.w2 = w2;
    P.on12.resize(n); P.on21.resize(n); P.chi12.assign(n, 0.); P.chi21.assign(n, 0.);
    for (int i = 0; i < n; i++) { P.on12[i] = !(skip12 && skip12[i]); P.on21[i] = !(skip21 && skip21[i]); status[i] = 0; }
    P.cam1 = {K1[0], K1[1], K1[2], K1[3]}; P.cam2 = {K2[0], K2[1], K2[2], K2[3]};
    P.fixScale = fixScale != 0; P.robust = robustFirst != 0;
    const float deltaHuber = std::sqrt(th2);
    P.delta = deltaHuber; P.dsqr = P.delta * P.delta;
    P.est = Sim3Problem::sim3_from(S8);
    lm_optimize(P, 5);
    auto put = [&](const Sim3 &S) { S8[0] = S.r.x; S8[1] = S.r.y; S8[2] = S.r.z; S8[3] = S.r.w; S8[4] = S.t.x; S8[5] = S.t.y; S8[6] = S.t.z; S8[7] = S.s; };
    put(P.est);                                                     // OptimizeCloudSim3 publishes the estimate here already (:2397)
    int nBad = 0;
    for (int i = 0; i < n; i++) {
        // chi2() of the errors the last computeActiveErrors() left on the edges (:2407 / :2110)
        if ((P.on12[i] && P.chi12[i] > (double)th2) || (P.on21[i] && P.chi21[i] > (double)th2)) { status[i] = 1; P.on12[i] = P.on21[i] = 0; nBad++; }
    }
    P.robust = false;
    res[0] = 0; res[1] = nBad; res[2] = 1;
    if (n - nBad < 10) return 0;
    lm_optimize(P, nBad > 0 ? 10 : 5);
    int nIn = 0;
    for (int i = 0; i < n; i++) {
        if (status[i] == 1) continue;
        if (!P.on12[i] || !P.on21[i]) { status[i] = 3; continue; }
        double e12[2], e21[2];
        P.errors(i, P.est, true, true, e12, e21);
        if (Sim3Problem::chi2_of(e12, w1[i]) > (double)th2 || Sim3Problem::chi2_of(e21, w2[i]) > (double)th2) status[i] = 2; else nIn++;
    }
    put(P.est);
    res[0] = nIn; res[2] = 0;
    return 0;
}

}  // extern "C"
