// Compiles rumi_slam_amd/facade/shells/*.cc -- the NON-template definitions with the reference's exact signatures -- against the reference's class
// declarations (tests/cpp/ref_decls/: signatures only, over the mock data model) and calls through them on the GPU: the same two frames through
// ORB_SLAM3::ORBmatcher (shell) and through the facade templates directly must give the same matches, and Optimizer::PoseOptimization (shell)
// the same pose as the template.
#define RUMI_HAVE_SOPHUS 1
#include <cstdio>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <random>
#include <vector>

#include "ORBmatcher.h"          // ref_decls: the reference's declaration
#include "Optimizer.h"           // ref_decls
#include "Sim3Solver.h"          // ref_decls
#include "ORBextractor.h"        // the facade extractor (replaces the reference's header and .cc as a whole)

// the shells themselves (a maintainer builds them as translation units of their own)
#include "../../rumi_slam_amd/facade/shells/ORBmatcher.cc"
#include "../../rumi_slam_amd/facade/shells/Optimizer_hot.cc"
#define RUMI_SHELLS_NO_EIGEN_GEOMETRY 1      // Eigen::umeyama is the reference's own dependency (not in this image)
#include "../../rumi_slam_amd/facade/shells/Sim3Solver.cc"

static int fails = 0;
#define CHECK(c, msg) do { if (!(c)) { std::printf("FAIL: %s (%s:%d)\n", msg, __FILE__, __LINE__); fails++; } } while (0)

int main(int argc, char **argv) {
    if (argc < 3) { std::printf("usage: test_shells frame0.bin frame1.bin (640x480 u8)\n"); return 2; }
    std::vector<uint8_t> im[2];
    for (int k = 0; k < 2; k++) {
        im[k].resize(640 * 480);
        FILE *f = std::fopen(argv[1 + k], "rb");
        if (!f || std::fread(im[k].data(), 1, im[k].size(), f) != im[k].size()) { std::printf("cannot read %s\n", argv[1 + k]); return 2; }
        std::fclose(f);
    }
    ORB_SLAM3::ORBextractor ext(1000, 1.2f, 8, 20, 7);
    std::vector<int> lap = {0, 1000};
    ORB_SLAM3::Frame fr[2];
    for (int k = 0; k < 2; k++) {
        cv::Mat image(480, 640, CV_8U, im[k].data(), 640), mask;
        ext(image, mask, fr[k].mvKeysUn, fr[k].mDescriptors, lap);
        fr[k].N = (int)fr[k].mvKeysUn.size();
        fr[k].mvScaleFactors = ext.GetScaleFactors(); fr[k].mvInvLevelSigma2 = ext.GetInverseScaleSigmaSquares(); fr[k].mvLevelSigma2 = ext.GetScaleSigmaSquares();
        fr[k].mvpMapPoints.assign(fr[k].N, nullptr); fr[k].mvbOutlier.assign(fr[k].N, false);
    }
    CHECK(fr[0].N > 900 && fr[1].N > 900, "extractor facade");
    // SearchForInitialization through the shell and through the template
    std::vector<cv::Point2f> prevA, prevB;
    for (auto &k : fr[0].mvKeysUn) { prevA.push_back(k.pt); prevB.push_back(k.pt); }
    std::vector<int> m12A, m12B;
    ORB_SLAM3::ORBmatcher shell(0.9f, true);
    rumi_facade_impl::ORBmatcher direct(0.9f, true);
    const int nA = shell.SearchForInitialization(fr[0], fr[1], prevA, m12A, 100);
    const int nB = direct.SearchForInitialization(fr[0], fr[1], prevB, m12B, 100);
    CHECK(nA > 50 && nA == nB && m12A == m12B, "ORBmatcher::SearchForInitialization: shell == template");
    CHECK(ORB_SLAM3::ORBmatcher::TH_HIGH == 100 && ORB_SLAM3::ORBmatcher::TH_LOW == 50 && ORB_SLAM3::ORBmatcher::HISTO_LENGTH == 30, "constants");
    CHECK(ORB_SLAM3::ORBmatcher::DescriptorDistance(fr[0].mDescriptors, fr[0].mDescriptors) == 0, "DescriptorDistance");
    // PoseOptimization through the shell: map points = back-projections of frame 0's key-points at 4 m, observed by frame 0 itself
    std::vector<ORB_SLAM3::MapPoint> mps(fr[0].N);
    for (int i = 0; i < fr[0].N; i++) {
        const float z = 4.f;
        mps[i].pos = Eigen::Vector3f((fr[0].mvKeysUn[i].pt.x - fr[0].cx) * z / fr[0].fx, (fr[0].mvKeysUn[i].pt.y - fr[0].cy) * z / fr[0].fy, z);
        fr[0].mvpMapPoints[i] = &mps[i];
    }
    fr[0].pose = Sophus::SE3f(Eigen::Quaternionf(1, 0, 0, 0), Eigen::Vector3f(0.02f, -0.01f, 0.03f));
    ORB_SLAM3::Frame copy = fr[0];
    const int gA = ORB_SLAM3::Optimizer::PoseOptimization(&fr[0]);
    const int gB = rumi_facade_impl::Optimizer::PoseOptimization(&copy);
    const Eigen::Vector3f tA = fr[0].GetPose().translation(), tB = copy.GetPose().translation();
    CHECK(gA == gB && gA > fr[0].N * 9 / 10 && tA(0) == tB(0) && tA(1) == tB(1) && tA(2) == tB(2), "Optimizer::PoseOptimization: shell == template");
    CHECK(std::fabs(tA(0)) < 1e-3f && std::fabs(tA(1)) < 1e-3f && std::fabs(tA(2)) < 1e-3f, "PoseOptimization finds the identity pose");
    // ---- Sim3Solver shell: blocks of iterations drawn ahead == upstream's one-at-a-time loop, and rand() left where that loop leaves it ----
    {
        std::mt19937 rng(11);
        std::uniform_real_distribution<float> U(-1.f, 1.f);
        const int NP = 120;
        ORB_SLAM3::KeyFrame A, B;
        A.mnId = 1; B.mnId = 2;
        A.pose = Sophus::SE3f(Eigen::Quaternionf(1, 0, 0, 0), Eigen::Vector3f(0.1f, 0.f, 0.2f));
        B.pose = Sophus::SE3f(Eigen::Quaternionf(1, 0, 0, 0), Eigen::Vector3f(-0.2f, 0.1f, 0.f));
        A.mvLevelSigma2 = ext.GetScaleSigmaSquares(); B.mvLevelSigma2 = A.mvLevelSigma2;
        std::vector<ORB_SLAM3::MapPoint> pa(NP), pb(NP);
        std::vector<ORB_SLAM3::MapPoint *> matched(NP, nullptr);
        const float sc = 1.3f;
        for (int i = 0; i < NP; i++) {
            const Eigen::Vector3f Xa(2.f * U(rng), 1.5f * U(rng), 4.f + 2.f * U(rng));        // in A's camera frame
            const Eigen::Vector3f Xb = (Xa - Eigen::Vector3f(0.3f, -0.1f, 0.2f)) / sc;         // B's camera frame: Xa = sc * Xb + t (R = I)
            Eigen::Vector3f nb = Xb;
            if (i % 7 == 0) nb = Xb + Eigen::Vector3f(0.6f * U(rng), 0.6f * U(rng), 0.6f * U(rng));   // gross outliers
            pa[i].pos = A.pose.inverse() * Xa; pb[i].pos = B.pose.inverse() * nb;
            cv::KeyPoint ka, kb; ka.octave = i % 4; kb.octave = (i + 1) % 4;
            ka.pt.x = A.cam.fx * Xa(0) / Xa(2) + A.cam.cx; ka.pt.y = A.cam.fy * Xa(1) / Xa(2) + A.cam.cy;
            kb.pt.x = B.cam.fx * nb(0) / nb(2) + B.cam.cx; kb.pt.y = B.cam.fy * nb(1) / nb(2) + B.cam.cy;
            A.mvKeysUn.push_back(ka); A.mvKeys.push_back(ka); B.mvKeysUn.push_back(kb); B.mvKeys.push_back(kb);
            A.mvpMapPoints.push_back(&pa[i]); B.mvpMapPoints.push_back(&pb[i]);
            pa[i].obs[&A] = std::make_tuple(i, -1); pb[i].obs[&B] = std::make_tuple(i, -1);
            if (i % 11 != 0) matched[i] = &pb[i];
        }
        A.N = B.N = NP;
        // (a) the shell: one block of 40 iterations
        srand(1234);
        ORB_SLAM3::Sim3Solver solver(&A, &B, matched, false);
        solver.SetRansacParameters(0.99, 60, 300);
        bool noMoreA = false; std::vector<bool> inlA; int nInA = 0;
        const Eigen::Matrix4f TA = solver.iterate(40, noMoreA, inlA, nInA);
        const int nextA = rand();
        // (b) upstream's loop, one iteration at a time with rand() itself: a solver per iteration block of 1 gives the same state machine
        srand(1234);
        ORB_SLAM3::Sim3Solver serial(&A, &B, matched, false);
        serial.SetRansacParameters(0.99, 60, 300);
        bool noMoreB = false; std::vector<bool> inlB; int nInB = 0;
        Eigen::Matrix4f TB = Eigen::Matrix4f::Identity();
        for (int it = 0; it < 40 && !noMoreB; it++) {
            TB = serial.iterate(1, noMoreB, inlB, nInB);
            if (nInB > 0) break;
        }
        const int nextB = rand();
        bool sameT = true;
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) sameT &= TA(r, c) == TB(r, c);
        CHECK(nInA > 60 && nInA == nInB && inlA == inlB && sameT, "Sim3Solver::iterate: a block of 40 == 40 blocks of 1");
        CHECK(nextA == nextB, "Sim3Solver::iterate leaves rand() where the one-at-a-time loop leaves it");
        CHECK(std::fabs(solver.GetEstimatedScale() - sc) < 0.02f && std::fabs(solver.GetEstimatedTranslation()(0) - 0.3f) < 0.05f, "Sim3Solver recovers the similarity");
        std::printf("Sim3Solver shell: %d inliers of %d kept correspondences, scale %.4f, next rand %d / %d\n", nInA, (int)std::count(matched.begin(), matched.end(), (ORB_SLAM3::MapPoint *)nullptr) * 0 + NP, solver.GetEstimatedScale(), nextA, nextB);
    }
    std::printf("shell test: %d failure(s); init matches %d, pose inliers %d\n", fails, nA, gA);
    return fails ? 1 : 0;
}
