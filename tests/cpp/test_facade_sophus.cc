// Compiles the RUMI_HAVE_SOPHUS sections of rumi-slam_amd/facade/*.h (the overloads that take Sophus::Sim3f / write back through
// Sophus::SE3f) against tests/cpp/mock_sophus.h and runs them on the GPU: SearchByProjection(KF, Sim3, ...) x 2, Fuse(KF, Sim3, ...),
// SearchBySim3, SearchForTriangulation with the fundamental matrix formed by the facade's own expression, PoseOptimization's SetPose.
// The numeric parity of the C-ABI calls underneath is covered by tests/test_matcher_gpu.py; here the results are checked against
// the same calls made directly with the arrays this test gathers itself.
#define RUMI_HAVE_SOPHUS 1
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <random>
#include <set>
#include <tuple>
#include <vector>

#include "mock_sophus.h"

#include "ORBextractor.h"
#include "ORBmatcher.h"
#include "Optimizer.h"
#include "Sim3Scoring.h"

#include "mock_model_sophus.h"

static int fails = 0;
#define CHECK(c, msg) do { if (!(c)) { std::printf("FAIL: %s (%s:%d)\n", msg, __FILE__, __LINE__); fails++; } } while (0)

int main(int argc, char **argv) {
    if (argc < 3) { std::printf("usage: test_facade_sophus frame0.bin frame1.bin (640x480 u8)\n"); return 2; }
    std::vector<uint8_t> im[2];
    for (int k = 0; k < 2; k++) {
        im[k].resize(640 * 480);
        FILE *f = std::fopen(argv[1 + k], "rb");
        if (!f || std::fread(im[k].data(), 1, im[k].size(), f) != im[k].size()) { std::printf("cannot read %s\n", argv[1 + k]); return 2; }
        std::fclose(f);
    }
    ORB_SLAM3::ORBextractor ext(1000, 1.2f, 8, 20, 7);
    std::vector<int> lap = {0, 1000};
    KeyFrame kf[2];
    for (int k = 0; k < 2; k++) {
        cv::Mat img(480, 640, CV_8UC1, im[k].data(), 640), mask;
        ext(img, mask, kf[k].mvKeysUn, kf[k].mDescriptors, lap);
        kf[k].N = (int)kf[k].mvKeysUn.size();
        kf[k].mvScaleFactors = ext.GetScaleFactors(); kf[k].mvInvLevelSigma2 = ext.GetInverseScaleSigmaSquares();
        kf[k].mvpMapPoints.assign(kf[k].N, nullptr); kf[k].mvbOutlier.assign(kf[k].N, false);
        kf[k].mnId = k;
        for (int i = 0; i < kf[k].N; i++) kf[k].mFeatVec[(unsigned)(i % 89)].push_back((unsigned)i);
    }
    kf[1].pose = Sophus::SE3f(Eigen::Quaternionf(0.99998f, 0.004f, -0.003f, 0.002f), Eigen::Vector3f(0.02f, -0.01f, 0.03f));
    std::mt19937 rng(3);
    // one map point per key-point of kf[0], placed so that it projects near the same pixel in kf[1]
    std::vector<MapPoint> mps(kf[0].N);
    std::vector<MapPoint *> vp;
    for (int i = 0; i < kf[0].N; i++) {
        const float z = 2.f + (rng() % 600) / 100.f;
        const Eigen::Vector3f pc((kf[0].mvKeysUn[i].pt.x - 320.1f) / 535.4f * z, (kf[0].mvKeysUn[i].pt.y - 247.6f) / 539.2f * z, z);
        mps[i].pos = kf[1].pose.inverse() * pc;
        mps[i].desc = cv::Mat(1, 32, CV_8U, kf[0].mDescriptors.ptr(i), 32);
        // UpdateNormalAndDepth: mfMaxDistance = dist * scale[level], mfMinDistance = mfMaxDistance / scale[nLevels - 1]
        const float dist = std::sqrt(pc(0) * pc(0) + pc(1) * pc(1) + pc(2) * pc(2));
        mps[i].maxD = dist * kf[0].mvScaleFactors[kf[0].mvKeysUn[i].octave]; mps[i].minD = mps[i].maxD / kf[0].mvScaleFactors[7];
        vp.push_back(&mps[i]);
    }
    ORB_SLAM3::ORBmatcher matcher(0.75f, true);
    // --- SearchByProjection(KF, Sim3, points, matched, th, ratio) and its pointKFs overload: same search, same result ---
    Sophus::Sim3f Scw(kf[1].pose.rotationMatrix(), kf[1].pose.translation(), 1.0f);
    {
        KeyFrame target = kf[0];                                   // features of frame 0 seen from pose kf[1].pose
        std::vector<MapPoint *> matched(target.N, nullptr), matched2(target.N, nullptr);
        std::vector<KeyFrame *> ptKFs(vp.size(), &kf[1]), matchedKF(target.N, nullptr);
        const int n1 = matcher.SearchByProjection(&target, Scw, vp, matched, 6, 1.0f);
        const int n2 = matcher.SearchByProjection(&target, Scw, vp, ptKFs, matched2, matchedKF, 6, 1.0f);
        int cnt = 0, self = 0;
        for (int f = 0; f < target.N; f++) { cnt += matched[f] != nullptr; self += matched[f] == &mps[f]; }
        CHECK(n1 > 300 && n1 == cnt, "SearchByProjection(KF, Sim3) count == filled slots");
        CHECK(self > n1 * 8 / 10, "SearchByProjection(KF, Sim3): a point built from feature f is matched back to feature f");
        CHECK(n2 > 300, "SearchByProjection(KF, Sim3, pointKFs) runs");
        bool kfok = true;
        for (int f = 0; f < target.N; f++) if ((matched2[f] != nullptr) != (matchedKF[f] == &kf[1])) kfok = false;
        CHECK(kfok, "matched key-frames follow matched points");
        // --- Fuse(KF, Sim3, points, th, replace): empty key-frame -> every fused point becomes an observation ---
        KeyFrame empty = kf[0];
        std::vector<MapPoint *> repl(vp.size(), nullptr);
        const int nf = matcher.Fuse(&empty, Scw, vp, 4.f, repl);
        int added = 0, replaced = 0;
        for (auto &p : mps) added += p.IsInKeyFrame(&empty);
        for (auto *r : repl) replaced += r != nullptr;
        CHECK(nf > 300 && nf == added + replaced && added > replaced, "Fuse(KF, Sim3): fused == observations added + duplicates reported");
        bool dupOk = true;                                         // a reported duplicate is the point that took the feature first
        for (size_t i = 0; i < repl.size(); i++) if (repl[i] && !repl[i]->IsInKeyFrame(&empty)) dupOk = false;
        CHECK(dupOk, "Fuse(KF, Sim3): vpReplacePoint holds the key-frame's point");
        std::vector<MapPoint *> repl2(vp.size(), nullptr);
        const int nf2 = matcher.Fuse(&empty, Scw, vp, 4.f, repl2);
        CHECK(nf2 == replaced, "Fuse(KF, Sim3) skips the points already in the key-frame and reports the same duplicates again");
    }
    // --- SearchBySim3: kf A (frame 0, pose I) and kf B (frame 0 again, pose P): S12 = P^-1 maps camera B into camera A ---
    {
        KeyFrame A = kf[0], B = kf[0];
        B.pose = kf[1].pose; B.mnId = 7;
        std::vector<MapPoint> ma(A.N), mb(B.N);
        for (int i = 0; i < A.N; i++) {
            const float z = 3.f + (i % 40) / 10.f;
            const Eigen::Vector3f pcA((A.mvKeysUn[i].pt.x - 320.1f) / 535.4f * z, (A.mvKeysUn[i].pt.y - 247.6f) / 539.2f * z, z);
            const float dist = std::sqrt(pcA(0) * pcA(0) + pcA(1) * pcA(1) + pcA(2) * pcA(2));
            ma[i].pos = pcA; ma[i].desc = cv::Mat(1, 32, CV_8U, A.mDescriptors.ptr(i), 32);
            ma[i].maxD = dist * A.mvScaleFactors[A.mvKeysUn[i].octave]; ma[i].minD = ma[i].maxD / A.mvScaleFactors[7];
            mb[i].pos = B.pose.inverse() * pcA;                    // the same point expressed so that camera B sees it at the same pixel
            mb[i].desc = ma[i].desc; mb[i].minD = ma[i].minD; mb[i].maxD = ma[i].maxD;
            A.mvpMapPoints[i] = &ma[i]; B.mvpMapPoints[i] = &mb[i];
        }
        const Sophus::Sim3f S12;                                   // identity: camera-A coordinates of a's points == camera-B coordinates of b's
        std::vector<MapPoint *> m12(A.N, nullptr);
        const int nfound = matcher.SearchBySim3(&A, &B, m12, S12, 7.5f);
        int self = 0;
        for (int i = 0; i < A.N; i++) self += m12[i] == &mb[i];
        CHECK(nfound > 500 && self > nfound * 9 / 10, "SearchBySim3 finds the mutual matches");
    }
    // --- SearchForTriangulation with F12 from the facade's own expression: for a pure translation between the two cameras a
    //     correspondence satisfies the epipolar constraint, a mismatched row does not
    {
        KeyFrame A = kf[0], B = kf[0];
        B.pose = Sophus::SE3f(Eigen::Quaternionf(1, 0, 0, 0), Eigen::Vector3f(0.3f, 0.f, 0.f));
        std::vector<std::pair<size_t, size_t>> pairs, coarse;
        const int nt = matcher.SearchForTriangulation(&A, &B, pairs, false, false);
        const int nc = matcher.SearchForTriangulation(&A, &B, coarse, false, true);
        CHECK(nt > 100 && (int)pairs.size() == nt && nc >= nt, "SearchForTriangulation (F12 from Sophus expressions) runs");
        int sameRow = 0;
        for (auto &p : pairs) sameRow += std::fabs(A.mvKeysUn[p.first].pt.y - B.mvKeysUn[p.second].pt.y) < 4.f * A.mvScaleFactors[B.mvKeysUn[p.second].octave];
        CHECK(sameRow == nt, "x-translation: accepted pairs lie on the same image row");
    }
    // --- PoseOptimization writes back through SetPose(Sophus::SE3f(Eigen::Quaternionf, Eigen::Vector3f)) ---
    {
        Frame F = kf[0];
        F.pose = Sophus::SE3f(Eigen::Quaternionf(0.99995f, 0.006f, -0.004f, 0.003f), Eigen::Vector3f(0.05f, -0.03f, 0.06f));
        for (int i = 0; i < F.N; i++) F.mvpMapPoints[i] = &mps[i];
        const int ng = ORB_SLAM3::Optimizer::PoseOptimization(&F);
        const Sophus::SE3f T = F.GetPose();
        CHECK(ng > 500, "PoseOptimization inliers");
        CHECK(std::fabs(T.translation()(0) - 0.02f) < 5e-3f && std::fabs(T.translation()(2) - 0.03f) < 5e-3f && std::fabs(T.unit_quaternion().x() - 0.004f) < 2e-3f,
              "PoseOptimization recovers the pose the points were built with");
    }
    // --- ComputeInliersNum: two key-frames observing the same points; world 2 = world 1 under a similarity -> all inliers ---
    {
        KeyFrame A = kf[0], B = kf[0];
        A.mvKeys = A.mvKeysUn; B.mvKeys = B.mvKeysUn;
        A.mvLevelSigma2.assign(8, 1.f); B.mvLevelSigma2.assign(8, 1.f);
        for (int l = 1; l < 8; l++) { A.mvLevelSigma2[l] = A.mvScaleFactors[l] * A.mvScaleFactors[l]; B.mvLevelSigma2[l] = A.mvLevelSigma2[l]; }
        const double sc = 2.0;
        g2o::Sim3 Sw1w2(Eigen::Quaterniond{1, 0, 0, 0}, Eigen::Vector3d{0.5, -0.25, 1.0}, sc);     // Pw1 = sc * Pw2 + t
        B.pose = Sophus::SE3f(Eigen::Quaternionf(1, 0, 0, 0), Eigen::Vector3f(0.25f, -0.125f, 0.5f));   // camera 2 = camera 1 seen from world 2: t / sc
        std::vector<MapPoint> ma(A.N), mb(B.N);
        std::vector<std::pair<int, int>> matches;
        for (int i = 0; i < A.N; i++) {
            const float z = 3.f + (i % 40) / 10.f;
            const Eigen::Vector3f pcA((A.mvKeysUn[i].pt.x - 320.1f) / 535.4f * z, (A.mvKeysUn[i].pt.y - 247.6f) / 539.2f * z, z);
            ma[i].pos = pcA;                                                                         // camera 1 at the origin of world 1
            mb[i].pos = Eigen::Vector3f((pcA(0) - 0.5f) / 2.f, (pcA(1) + 0.25f) / 2.f, (pcA(2) - 1.0f) / 2.f);
            if (i % 9 == 0) mb[i].pos = mb[i].pos + Eigen::Vector3f(0.3f, 0.f, 0.f);                   // a few gross outliers
            A.mvpMapPoints[i] = &ma[i]; B.mvpMapPoints[i] = &mb[i];
            matches.push_back({i, i});
        }
        std::vector<KeyFrame *> m1 = {&A, nullptr}, m2 = {&B, &B};
        std::vector<std::vector<std::pair<int, int>>> all = {matches, matches};
        const float ratio = rumi_facade::ComputeInliersNum(m1, m2, all, Sw1w2);
        CHECK(ratio > 0.85f && ratio < 0.92f, "ComputeInliersNum: 8/9 of the matches are inliers");
        std::printf("ComputeInliersNum ratio %.4f\n", ratio);
    }
    // --- OptimizeSim3 / OptimizeCloudSim3: key-frame B sees the points of key-frame A from a world that is a similarity away ---
    {
        KeyFrame A = kf[0], B = kf[0];
        const double sc = 1.25;
        // S12 maps camera-2 coordinates into camera-1 coordinates: P1c = sc * P2c + t (no rotation)
        std::vector<MapPoint> ma(A.N), mb(B.N);
        std::vector<MapPoint *> matches(A.N, nullptr);
        A.pose = Sophus::SE3f(); B.pose = Sophus::SE3f();
        for (int i = 0; i < A.N; i++) {
            const float z = 3.f + (i % 40) / 10.f;
            const Eigen::Vector3f pcA((A.mvKeysUn[i].pt.x - 320.1f) / 535.4f * z, (A.mvKeysUn[i].pt.y - 247.6f) / 539.2f * z, z);
            ma[i].pos = pcA;
            mb[i].pos = Eigen::Vector3f((pcA(0) - 0.1f) / (float)sc, (pcA(1) + 0.05f) / (float)sc, (pcA(2) - 0.2f) / (float)sc);
            A.mvpMapPoints[i] = &ma[i];
            // key-frame 2 observes its point where it projects (feature i of B), except for a few gross outliers
            const Eigen::Vector2f uv = B.cam.project(mb[i].pos);
            B.mvKeysUn[i].pt.x = uv(0) + (i % 11 == 0 ? 40.f : 0.f); B.mvKeysUn[i].pt.y = uv(1);
            mb[i].obs[&B] = std::make_tuple(i, -1);
            if (i % 3) matches[i] = &mb[i];
        }
        g2o::Sim3 S12(Eigen::Quaterniond{1, 0, 0, 0}, Eigen::Vector3d{0.12, -0.04, 0.22}, 1.22);       // perturbed start
        Eigen::Matrix77d H;
        std::vector<MapPoint *> m1 = matches;
        int nGiven = 0, nOut = 0;
        for (int i = 0; i < A.N; i++) if (matches[i]) { nGiven++; nOut += (i % 11 == 0); }
        const int nIn = ORB_SLAM3::Optimizer::OptimizeSim3(&A, &B, m1, S12, 10.f, false, H, true);
        int nNulled = 0, outNulled = 0;
        for (int i = 0; i < A.N; i++) if (matches[i] && !m1[i]) { nNulled++; outNulled += (i % 11 == 0); }
        std::printf("OptimizeSim3: %d given, %d inliers, %d removed (%d of %d outliers), s %.5f t %.4f %.4f %.4f\n", nGiven, nIn, nNulled, outNulled, nOut, S12.scale(),
                    S12.translation()(0), S12.translation()(1), S12.translation()(2));
        CHECK(nIn == nGiven - nNulled && outNulled == nOut && nNulled <= nOut + nGiven / 50, "OptimizeSim3 removes exactly the gross outliers");
        CHECK(std::fabs(S12.scale() - sc) < 2e-3 && std::fabs(S12.translation()(0) - 0.1) < 3e-3 && std::fabs(S12.translation()(2) - 0.2) < 6e-3, "OptimizeSim3 recovers the similarity");
        CHECK(H.m[3][3] == 0.0, "mAcumHessian zeroed");
        // the cloud variant over two key-frame pairs (the same pair twice), fixed scale: the vertex is gSw1w2 and both key-frames sit at their worlds' origins
        g2o::Sim3 Sw(Eigen::Quaterniond{1, 0, 0, 0}, Eigen::Vector3d{0.11, -0.05, 0.21}, sc);
        std::vector<KeyFrame *> k1 = {&A, &A}, k2 = {&B, &B};
        std::vector<std::vector<MapPoint *>> av = {matches, matches};
        ma[5].isEdge = true;
        const float ratio = ORB_SLAM3::Optimizer::OptimizeCloudSim3(k1, k2, av, Sw, 10.f, true, H, true);
        std::printf("OptimizeCloudSim3: ratio %.4f, t %.4f %.4f %.4f s %.4f\n", ratio, Sw.translation()(0), Sw.translation()(1), Sw.translation()(2), Sw.scale());
        CHECK(ratio > 0.85f && ratio < 0.93f, "OptimizeCloudSim3 inlier ratio (1/11 of the matches are outliers)");
        CHECK(Sw.scale() == sc && std::fabs(Sw.translation()(0) - 0.1) < 3e-3 && std::fabs(Sw.translation()(2) - 0.2) < 6e-3, "OptimizeCloudSim3 recovers the translation, keeps the scale");
    }
    // --- GlobalBundleAdjustemnt / BundleAdjustment: six key-frames around the points of kf[0], perturbed poses and points ---
    {
        Map M;
        const int NK = 6, NP = std::min(kf[0].N, 600);
        std::vector<KeyFrame> ks(NK);
        std::vector<MapPoint> ps(NP);
        std::vector<Sophus::SE3f> truth(NK);
        for (int k = 0; k < NK; k++) {
            ks[k].mnId = k; ks[k].map = &M; ks[k].mvInvLevelSigma2 = kf[0].mvInvLevelSigma2;
            truth[k] = Sophus::SE3f(Eigen::Quaternionf(1.f, 0.f, 0.004f * k, 0.f), Eigen::Vector3f(-0.05f * k, 0.01f * k, 0.02f * k));
            M.allKFs.push_back(&ks[k]);
        }
        for (int i = 0; i < NP; i++) {
            const float z = 2.f + (i % 50) / 10.f;
            const Eigen::Vector3f X((kf[0].mvKeysUn[i].pt.x - 320.1f) / 535.4f * z, (kf[0].mvKeysUn[i].pt.y - 247.6f) / 539.2f * z, z);
            ps[i].pos = X + Eigen::Vector3f(0.01f * ((i * 7) % 5 - 2), 0.01f * ((i * 3) % 5 - 2), 0.02f * ((i * 11) % 5 - 2));   // perturbed map point
            ps[i].map = &M;
            for (int k = 0; k < NK; k++) {
                const Eigen::Vector2f uv = ks[k].cam.project(truth[k] * X);
                if (uv(0) < 0 || uv(0) >= 640 || uv(1) < 0 || uv(1) >= 480) continue;
                cv::KeyPoint kp = kf[0].mvKeysUn[i];
                kp.pt.x = uv(0); kp.pt.y = uv(1);
                ps[i].obs[&ks[k]] = std::make_tuple((int)ks[k].mvKeysUn.size(), -1);
                ks[k].mvKeysUn.push_back(kp); ks[k].mvuRight.push_back(-1.f); ks[k].mvpMapPoints.push_back(&ps[i]);
            }
            M.allMPs.push_back(&ps[i]);
        }
        for (int k = 0; k < NK; k++) {
            ks[k].N = (int)ks[k].mvKeysUn.size();
            ks[k].pose = k == 0 ? truth[0] : Sophus::SE3f(truth[k].unit_quaternion(), truth[k].translation() + Eigen::Vector3f(0.01f, -0.008f, 0.012f));
        }
        auto reproj = [&](bool gba) {
            double sum = 0; int cnt = 0;
            for (int i = 0; i < NP; i++)
                for (auto &ob : ps[i].obs) {
                    KeyFrame *k = ob.first;
                    const Eigen::Vector2f uv = k->cam.project((gba ? k->mTcwGBA : k->pose) * (gba ? ps[i].mPosGBA : ps[i].pos));
                    const auto &kp = k->mvKeysUn[std::get<0>(ob.second)];
                    sum += std::hypot(uv(0) - kp.pt.x, uv(1) - kp.pt.y); cnt++;
                }
            return sum / cnt;
        };
        const double e0 = reproj(false);
        const Sophus::SE3f pose1 = ks[1].pose;
        ORB_SLAM3::Optimizer::GlobalBundleAdjustemnt(&M, 10, nullptr, 7, false);            // nLoopKF != origin: results parked in mTcwGBA / mPosGBA
        const double eG = reproj(true);
        CHECK(ks[1].pose.translation()(0) == pose1.translation()(0) && ks[1].mnBAGlobalForKF == 7 && ps[3].mnBAGlobalForKF == 7, "GlobalBundleAdjustemnt parks its result when nLoopKF is not the origin");
        ORB_SLAM3::Optimizer::GlobalBundleAdjustemnt(&M, 10, nullptr, 0, false);            // nLoopKF == origin: SetPose / SetWorldPos
        const double e1 = reproj(false);
        std::printf("GlobalBundleAdjustemnt: mean reprojection error %.3f px -> %.4f px (parked result %.4f px)\n", e0, e1, eG);
        CHECK(e0 > 1.0 && e1 < 0.05 * e0 && std::fabs(eG - e1) < 1e-3, "GlobalBundleAdjustemnt reduces the reprojection error");
        CHECK(ks[0].pose.translation()(0) == truth[0].translation()(0), "the map's first key-frame stays fixed");
    }
    if (fails == 0) std::printf("facade (Sophus overloads): all checks passed\n");
    return fails ? 1 : 0;
}
