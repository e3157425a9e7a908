// Replaces R/lib_src/Sim3Solver.cc.  The reference's own header (include/cloud_edge_slam_lib/Sim3Solver.h) stays, and so do its data
// members: the constructor gathers the correspondences exactly as upstream (:40-132), the three iterate() overloads keep the RANSAC state
// machine on the host, and every call evaluates its whole block of iterations in ONE launch (include/rumi_opt.h: rumi_sim3_ransac -- one
// workgroup per hypothesis: Horn's closed form, CheckInliers, and for the rumination overload ComputeInliersNum over every key-frame pair).
//
// Minimal sets.  Upstream draws them with DUtils::Random::RandomInt = glibc rand() (Thirdparty/DBoW2/DUtils/Random.cpp:47-50) one iteration at a
// time and stops drawing when an iteration converges.  The draws do not depend on the hypotheses, so a block is drawn AHEAD on a private copy of the
// process generator's state (glibc's own random_r on a copy of its state array); once the replay of the block knows how many iterations upstream
// would have run, the real generator is advanced by exactly that many rand() calls -- rand() is left where upstream's loop would have left it.
//
// Build instead of lib_src/Sim3Solver.cc with -DRUMI_HAVE_SOPHUS -DRUMI_HAVE_OPENCV, link librumi_hip.so.  Not thread-safe with respect to other
// threads calling rand() during the two lines that copy the generator state (upstream's use of rand() from several threads is not either).
#include "Sim3Solver.h"          // the REFERENCE's header

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "KeyFrame.h"
#include "MapPoint.h"

#ifndef RUMI_FACADE_NAMESPACE
#define RUMI_FACADE_NAMESPACE rumi_facade_impl
#endif
#include "../Sim3Scoring.h"      // rumi_facade::ComputeInliersNum + the optimiser arena (this repository)

namespace ORB_SLAM3 {

namespace {
// glibc's process-wide generator, copied: random_r on the copy gives the values rand() is about to give
struct RandLookahead {
    int32_t table[32];
    char scratch[128], dummy[128];
    struct random_data rd;
    RandLookahead() {
        char *prev = initstate(1u, scratch, sizeof scratch);      // the global generator moves onto `scratch`; glibc returns its own state array,
        std::memcpy(table, prev, sizeof table);                    // word 0 of which now encodes the position it had reached
        setstate(prev);                                            // ... and resumes exactly there
        std::memset(&rd, 0, sizeof rd);
        initstate_r(1u, dummy, sizeof dummy, &rd);
        setstate_r(reinterpret_cast<char *>(table), &rd);
    }
    int next() { int32_t v = 0; random_r(&rd, &v); return (int)v; }
    int RandomInt(int min, int max) { const int d = max - min + 1; return int(((double)next() / ((double)RAND_MAX + 1.0)) * d) + min; }   // DUtils/Random.cpp:47-50
};

inline size_t float_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return (size_t)u; }
inline float bits_float(size_t s) { const uint32_t u = (uint32_t)s; float f; std::memcpy(&f, &u, 4); return f; }

template <class M3, class V3> inline Eigen::Matrix4f make_T(const M3 &R, const V3 &t, float s) {   // ComputeSim3's mT12i (:527-532)
    Eigen::Matrix4f T = Eigen::Matrix4f::Identity();
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) T(r, c) = s * R(r, c); T(r, 3) = t(r); }
    return T;
}
template <class S> inline void push_sim3(std::vector<double> &v, const S &g) {
    const auto q = g.rotation(); const auto t = g.translation();
    const double a[8] = {q.x(), q.y(), q.z(), q.w(), t(0), t(1), t(2), g.scale()};
    v.insert(v.end(), a, a + 8);
}
}  // namespace

Sim3Solver::Sim3Solver() {}

Sim3Solver::Sim3Solver(KeyFrame *pKF1, KeyFrame *pKF2, const std::vector<MapPoint *> &vpMatched12, const bool bFixScale,
                       std::vector<KeyFrame *> vpKeyFrameMatchedMP)
    : mnIterations(0), mnBestInliers(0), mbFixScale(bFixScale), pCamera1(pKF1->mpCamera), pCamera2(pKF2->mpCamera) {
    bool bDifferentKFs = true;
    if (vpKeyFrameMatchedMP.empty()) { bDifferentKFs = false; vpKeyFrameMatchedMP = std::vector<KeyFrame *>(vpMatched12.size(), pKF2); }
    mpKF1 = pKF1; mpKF2 = pKF2;
    const std::vector<MapPoint *> vpKeyFrameMP1 = pKF1->GetMapPointMatches();
    mN1 = (int)vpMatched12.size();
    mvpMatches12 = vpMatched12;
    const Eigen::Matrix3f Rcw1 = pKF1->GetRotation(), Rcw2 = pKF2->GetRotation();
    const Eigen::Vector3f tcw1 = pKF1->GetTranslation(), tcw2 = pKF2->GetTranslation();
    size_t idx = 0;
    KeyFrame *pKFm = pKF2;
    for (int i1 = 0; i1 < mN1; i1++) {                               // :78-126
        if (!vpMatched12[i1]) continue;
        MapPoint *pMP1 = vpKeyFrameMP1[i1], *pMP2 = vpMatched12[i1];
        if (!pMP1) continue;
        if (pMP1->isBad() || pMP2->isBad()) continue;
        if (bDifferentKFs) pKFm = vpKeyFrameMatchedMP[i1];
        const int indexKF1 = std::get<0>(pMP1->GetIndexInKeyFrame(pKF1)), indexKF2 = std::get<0>(pMP2->GetIndexInKeyFrame(pKFm));
        if (indexKF1 < 0 || indexKF2 < 0) continue;
        const cv::KeyPoint &kp1 = pKF1->mvKeysUn[indexKF1];
        const cv::KeyPoint &kp2 = pKFm->mvKeysUn[indexKF2];
        const float sigmaSquare1 = pKF1->mvLevelSigma2[kp1.octave], sigmaSquare2 = pKFm->mvLevelSigma2[kp2.octave];
        mvnMaxError1.push_back(9.210 * sigmaSquare1);
        mvnMaxError2.push_back(9.210 * sigmaSquare2);
        mvSigmaSquare1.push_back(float_bits(sigmaSquare1));           // (declared by the reference, unused there: holds the float's bits for the device call)
        mvSigmaSquare2.push_back(float_bits(sigmaSquare2));
        mvpMapPoints1.push_back(pMP1); mvpMapPoints2.push_back(pMP2);
        mvnIndices1.push_back(i1);
        const Eigen::Vector3f X3D1w = pMP1->GetWorldPos(), X3D2w = pMP2->GetWorldPos();
        mvX3Dc1.push_back(Rcw1 * X3D1w + tcw1);
        mvX3Dc2.push_back(Rcw2 * X3D2w + tcw2);
        mvAllIndices.push_back(idx);
        idx++;
    }
    // mvP1im1 / mvP2im2 (:128-129) are re-projected on the device by every launch
    SetRansacParameters();
}

void Sim3Solver::SetRansacParameters(double probability, int minInliers, int maxIterations) {      // :134-157
    mRansacProb = probability; mRansacMinInliers = minInliers; mRansacMaxIts = maxIterations;
    N = (int)mvpMapPoints1.size();
    mvbInliersi.resize(N);
    const float epsilon = (float)mRansacMinInliers / N;
    int nIterations;
    if (mRansacMinInliers == N) nIterations = 1;
    else nIterations = (int)std::ceil(std::log(1 - mRansacProb) / std::log(1 - std::pow(epsilon, 3)));
    mRansacMaxIts = std::max(1, std::min(nIterations, mRansacMaxIts));
    mnIterations = 0;
}

namespace {
// One block of iterations: the minimal sets drawn ahead, one launch, per-hypothesis outputs
struct Block {
    int count = 0;
    std::vector<int32_t> tri, nIn;
    std::vector<float> T, median;
    std::vector<uint8_t> inl;
};
}  // namespace

// shared by the three overloads; `score` = NULL for the first two.  Returns false when the device call failed (reported through rumi_status.h).
static bool run_block(int count, int N, const std::vector<size_t> &all, const std::vector<Eigen::Vector3f> &X1v, const std::vector<Eigen::Vector3f> &X2v,
                      const std::vector<size_t> &s1bits, const std::vector<size_t> &s2bits, GeometricCamera *cam1, GeometricCamera *cam2, bool fixScale,
                      const RumiSim3ScoreSet *score, Block &B) {
    B.count = count;
    B.tri.resize((size_t)count * 3); B.nIn.assign(count, 0); B.T.assign((size_t)count * 16, 0.f); B.median.assign(count, 0.f);
    B.inl.assign((size_t)count * std::max(N, 1), 0);
    RandLookahead ahead;
    std::vector<size_t> avail;
    for (int h = 0; h < count; h++) {                                 // :176-191, on the copy of the generator
        avail = all;
        for (int i = 0; i < 3; i++) {
            const int randi = ahead.RandomInt(0, (int)avail.size() - 1);
            B.tri[(size_t)h * 3 + i] = (int32_t)avail[randi];
            avail[randi] = avail.back();
            avail.pop_back();
        }
    }
    std::vector<float> X1((size_t)N * 3), X2((size_t)N * 3), s1(N), s2(N);
    for (int i = 0; i < N; i++) {
        for (int c = 0; c < 3; c++) { X1[(size_t)i * 3 + c] = X1v[i](c); X2[(size_t)i * 3 + c] = X2v[i](c); }
        s1[i] = bits_float(s1bits[i]); s2[i] = bits_float(s2bits[i]);
    }
    const float K1[4] = {cam1->getParameter(0), cam1->getParameter(1), cam1->getParameter(2), cam1->getParameter(3)};
    const float K2[4] = {cam2->getParameter(0), cam2->getParameter(1), cam2->getParameter(2), cam2->getParameter(3)};
    return RUMI_GUARDED("Sim3Solver::iterate / rumi_sim3_ransac", &RUMI_FACADE_NAMESPACE::Optimizer::grow_arena,
                        rumi_sim3_ransac(RUMI_FACADE_NAMESPACE::Optimizer::arena(), N, X1.data(), X2.data(), s1.data(), s2.data(), K1, K2, fixScale ? 1 : 0, count,
                                         B.tri.data(), score, B.T.data(), B.nIn.data(), B.inl.data(), nullptr, B.median.data())) == RUMI_OK;
}

// iteration h of a block becomes the solver's "current estimation" (mR12i ... mnInliersi); a degenerate minimal set keeps the previous
// iteration's transform and inliers, as ComputeSim3's early return does (:493-494)
#define RUMI_TAKE_HYPOTHESIS(B, h)                                                                                  \
    do {                                                                                                            \
        const float *T_ = &(B).T[(size_t)(h) * 16];                                                                 \
        if (T_[13] != 0.f) {                                                                                        \
            for (int r_ = 0; r_ < 3; r_++) { for (int c_ = 0; c_ < 3; c_++) mR12i(r_, c_) = T_[r_ * 3 + c_]; mt12i(r_) = T_[9 + r_]; }   \
            ms12i = T_[12];                                                                                         \
            mT12i = make_T(mR12i, mt12i, ms12i);                                                                    \
            mnInliersi = (B).nIn[h];                                                                                \
            for (int i_ = 0; i_ < N; i_++) mvbInliersi[i_] = (B).inl[(size_t)(h) * N + i_] != 0;                    \
        }                                                                                                           \
    } while (0)
#define RUMI_TAKE_BEST()                                                                                            \
    do { mvbBestInliers = mvbInliersi; mnBestInliers = mnInliersi; mBestT12 = mT12i; mBestRotation = mR12i; mBestTranslation = mt12i; mBestScale = ms12i; } while (0)

Eigen::Matrix4f Sim3Solver::iterate(int nIterations, bool &bNoMore, std::vector<bool> &vbInliers, int &nInliers) {                    // :159-220
    bNoMore = false;
    vbInliers = std::vector<bool>(mN1, false);
    nInliers = 0;
    if (N < mRansacMinInliers) { bNoMore = true; return Eigen::Matrix4f::Identity(); }
    const int count = std::max(0, std::min(nIterations, mRansacMaxIts - mnIterations));
    Block B;
    if (count > 0 && !run_block(count, N, mvAllIndices, mvX3Dc1, mvX3Dc2, mvSigmaSquare1, mvSigmaSquare2, pCamera1, pCamera2, mbFixScale, nullptr, B)) {
        bNoMore = true;                                               // reported; no estimate from this solver
        return Eigen::Matrix4f::Identity();
    }
    for (int h = 0; h < count; h++) {
        mnIterations++;
        for (int k = 0; k < 3; k++) (void)rand();                     // the real generator follows, three draws per iteration upstream really runs
        RUMI_TAKE_HYPOTHESIS(B, h);
        if (mnInliersi >= mnBestInliers) {
            RUMI_TAKE_BEST();
            if (mnInliersi > mRansacMinInliers) {
                nInliers = mnInliersi;
                for (int i = 0; i < N; i++) if (mvbInliersi[i]) vbInliers[mvnIndices1[i]] = true;
                return mBestT12;
            }
        }
    }
    if (mnIterations >= mRansacMaxIts) bNoMore = true;
    return Eigen::Matrix4f::Identity();
}

Eigen::Matrix4f Sim3Solver::iterate(int nIterations, bool &bNoMore, std::vector<bool> &vbInliers, int &nInliers, bool &bConverge) {   // :222-290
    bNoMore = false; bConverge = false;
    vbInliers = std::vector<bool>(mN1, false);
    nInliers = 0;
    if (N < mRansacMinInliers) { bNoMore = true; return Eigen::Matrix4f::Identity(); }
    const int count = std::max(0, std::min(nIterations, mRansacMaxIts - mnIterations));
    Block B;
    if (count > 0 && !run_block(count, N, mvAllIndices, mvX3Dc1, mvX3Dc2, mvSigmaSquare1, mvSigmaSquare2, pCamera1, pCamera2, mbFixScale, nullptr, B)) {
        bNoMore = true;
        return Eigen::Matrix4f::Identity();
    }
    Eigen::Matrix4f bestSim3 = Eigen::Matrix4f::Identity();           // (upstream returns an uninitialised matrix when no iteration improved)
    for (int h = 0; h < count; h++) {
        mnIterations++;
        for (int k = 0; k < 3; k++) (void)rand();
        RUMI_TAKE_HYPOTHESIS(B, h);
        if (mnInliersi >= mnBestInliers) {
            RUMI_TAKE_BEST();
            nInliers = mnInliersi;
            if (mnInliersi > mRansacMinInliers) {
                for (int i = 0; i < N; i++) if (mvbInliersi[i]) vbInliers[mvnIndices1[i]] = true;
                bConverge = true;
                return mBestT12;
            }
            bestSim3 = mBestT12;
        }
    }
    if (mnIterations >= mRansacMaxIts) bNoMore = true;
    return bestSim3;
}

Eigen::Matrix4f Sim3Solver::iterate(int nIterations, bool &bNoMore, std::vector<bool> &vbInliers, int &nInliers, bool &bConverge,
                                    const std::vector<KeyFrame *> &Map1KFs, const std::vector<KeyFrame *> &Map2KFs,
                                    const std::vector<std::vector<std::pair<int, int>>> &avpValidKPMatches, float &bestRatio, Eigen::Matrix3f &bestRotation,
                                    Eigen::Vector3f &bestTranslation, float &bestScale) {                                             // :292-404
    bNoMore = false; bConverge = false;
    vbInliers = std::vector<bool>(mN1, false);
    nInliers = 0;
    if (N < mRansacMinInliers) { bNoMore = true; return Eigen::Matrix4f::Identity(); }
    // the data ComputeInliersNum reads (:586-664), gathered once per call; the composition with every hypothesis happens on the device
    std::vector<int32_t> pairStart(1, 0), denom;
    std::vector<double> Sc1w1, Sc2w2, Skf1, Skf2;
    std::vector<float> X1, X2, k1, k2, s1, s2;
    std::vector<uint8_t> e1, e2;
    float K1[4] = {0, 0, 0, 0}, K2[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < Map1KFs.size(); i++) {
        KeyFrame *KF1 = Map1KFs[i], *KF2 = Map2KFs[i];
        if (!KF1 || !KF2) continue;
        push_sim3(Sc1w1, g2o::Sim3(KF1->GetRotation().cast<double>(), KF1->GetTranslation().cast<double>(), 1.0));
        push_sim3(Sc2w2, g2o::Sim3(KF2->GetRotation().cast<double>(), KF2->GetTranslation().cast<double>(), 1.0));
        K1[0] = KF1->fx; K1[1] = KF1->fy; K1[2] = KF1->cx; K1[3] = KF1->cy;
        K2[0] = KF2->fx; K2[1] = KF2->fy; K2[2] = KF2->cx; K2[3] = KF2->cy;
        const std::vector<MapPoint *> vpMPs1 = KF1->GetMapPointMatches(), vpMPs2 = KF2->GetMapPointMatches();
        for (const auto &m : avpValidKPMatches[i]) {
            MapPoint *p1 = vpMPs1[m.first], *p2 = vpMPs2[m.second];
            if (!p1 || !p2) continue;
            const Eigen::Vector3f P1 = p1->GetWorldPos(), P2 = p2->GetWorldPos();
            for (int c = 0; c < 3; c++) { X1.push_back(P1(c)); X2.push_back(P2(c)); }
            const cv::KeyPoint &kp1 = KF1->mvKeys[m.first];
            const cv::KeyPoint &kp2 = KF2->mvKeys[m.second];
            k1.push_back(kp1.pt.x); k1.push_back(kp1.pt.y); k2.push_back(kp2.pt.x); k2.push_back(kp2.pt.y);
            s1.push_back(KF1->mvLevelSigma2[kp1.octave]); s2.push_back(KF2->mvLevelSigma2[kp2.octave]);
            e1.push_back(p1->isEdge); e2.push_back(p2->isEdge);
        }
        pairStart.push_back((int32_t)e1.size());
        denom.push_back((int32_t)avpValidKPMatches[i].size());
    }
    push_sim3(Skf1, g2o::Sim3(mpKF1->GetRotation().cast<double>(), mpKF1->GetTranslation().cast<double>(), 1.0));      // gSc1w (:342)
    push_sim3(Skf2, g2o::Sim3(mpKF2->GetRotation().cast<double>(), mpKF2->GetTranslation().cast<double>(), 1.0));      // gSc2w (:343)
    if (e1.empty()) { X1.resize(3); X2.resize(3); k1.resize(2); k2.resize(2); s1.resize(1); s2.resize(1); e1.resize(1); e2.resize(1); }
    RumiSim3ScoreSet S{(int32_t)denom.size(), pairStart.data(), denom.data(), Sc1w1.data(), Sc2w2.data(), Skf1.data(), Skf2.data(), K1, K2,
                       X1.data(), X2.data(), k1.data(), k2.data(), s1.data(), s2.data(), e1.data(), e2.data()};
    const int count = std::max(0, std::min(nIterations, mRansacMaxIts - mnIterations));
    Block B;
    if (count > 0 && !run_block(count, N, mvAllIndices, mvX3Dc1, mvX3Dc2, mvSigmaSquare1, mvSigmaSquare2, pCamera1, pCamera2, mbFixScale, denom.empty() ? nullptr : &S, B)) {
        bNoMore = true;
        return Eigen::Matrix4f::Identity();
    }
    Eigen::Matrix4f bestSim3 = Eigen::Matrix4f::Identity();
    for (int h = 0; h < count; h++) {
        mnIterations++;
        for (int k = 0; k < 3; k++) (void)rand();
        RUMI_TAKE_HYPOTHESIS(B, h);
        const float InliersRatio = denom.empty() ? 0.f : B.median[h];   // ComputeInliersNum returns 0 without key-frame pairs
        if (InliersRatio >= bestRatio && mnInliersi >= mnBestInliers) {
            RUMI_TAKE_BEST();
            bestRatio = InliersRatio;
            bestRotation = mR12i; bestTranslation = mt12i; bestScale = ms12i;
            if (InliersRatio > 0.10 && mnInliersi > mRansacMinInliers) { bConverge = true; return mBestT12; }
            bestSim3 = mBestT12;
        }
    }
    if (mnIterations >= mRansacMaxIts) bNoMore = true;
    return bestSim3;
}

Eigen::Matrix4f Sim3Solver::find(std::vector<bool> &vbInliers12, int &nInliers) {      // :425-428
    bool bFlag;
    return iterate(mRansacMaxIts, bFlag, vbInliers12, nInliers);
}

float Sim3Solver::ComputeInliersNum(const std::vector<KeyFrame *> &map1KFs, const std::vector<KeyFrame *> &map2KFs,
                                    const std::vector<std::vector<std::pair<int, int>>> &avpValidKPMatches, g2o::Sim3 &gSw1w2) {
    return rumi_facade::ComputeInliersNum(map1KFs, map2KFs, avpValidKPMatches, gSw1w2);
}

#ifndef RUMI_SHELLS_NO_EIGEN_GEOMETRY
// one call per merge, on the host as upstream (:406-423): Eigen::umeyama is the reference's own dependency
Eigen::Matrix4d Sim3Solver::umeyamaSolve(const std::vector<Eigen::Vector3d> &srcMatchPoints, const std::vector<Eigen::Vector3d> &dstMatchPoints) {
    Eigen::Matrix<double, 3, Eigen::Dynamic> src(3, srcMatchPoints.size()), dst(3, dstMatchPoints.size());
    for (size_t i = 0; i < srcMatchPoints.size(); i++)
        for (int c = 0; c < 3; c++) { src(c, i) = srcMatchPoints[i](c); dst(c, i) = dstMatchPoints[i](c); }
    return Eigen::umeyama(src, dst, true);
}
#endif

Eigen::Matrix4f Sim3Solver::GetEstimatedTransformation() { return mBestT12; }
Eigen::Matrix3f Sim3Solver::GetEstimatedRotation() { return mBestRotation; }
Eigen::Vector3f Sim3Solver::GetEstimatedTranslation() { return mBestTranslation; }
float Sim3Solver::GetEstimatedScale() { return mBestScale; }

#undef RUMI_TAKE_HYPOTHESIS
#undef RUMI_TAKE_BEST

}  // namespace ORB_SLAM3
