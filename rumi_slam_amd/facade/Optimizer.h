// Drop-in for the two hot members of the all-static ORB_SLAM3::Optimizer (R/include/cloud_edge_slam_lib/Optimizer.h:53,55)
// over the MI355X C ABI (include/rumi_opt.h).  Templates over the data-model types, same member names as the reference:
// graph gathering (Optimizer.cc:763-897, :1011-1271), locking (MapPoint::mGlobalMutex, Map::mMutexMapUpdate) and the
// write-back / observation erasing (:993-1000, :1325-1354) stay here on the host; the arithmetic runs on the GPU.
#pragma once
#include <cstdio>
#include <list>
#include <map>
#include <mutex>
#include <set>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "rumi_opt.h"
#include "rumi_status.h"

// capacity of the per-thread optimiser arenas (key-frames incl. fixed ones, map points, observations of one call)
#ifndef RUMI_OPT_MAX_KF
#define RUMI_OPT_MAX_KF 512
#endif
#ifndef RUMI_OPT_MAX_MP
#define RUMI_OPT_MAX_MP (1 << 17)
#endif
#ifndef RUMI_OPT_MAX_EDGES
#define RUMI_OPT_MAX_EDGES (1 << 20)
#endif

// The classes live in ORB_SLAM3 when these headers REPLACE the reference's (member templates deduce the reference's own types at every call
// site), or in a namespace of their own when the reference's headers stay and facade/shells/*.cc forward to them (-DRUMI_FACADE_NAMESPACE=...).
#ifndef RUMI_FACADE_NAMESPACE
#define RUMI_FACADE_NAMESPACE ORB_SLAM3
#endif
namespace RUMI_FACADE_NAMESPACE {

class Optimizer {
    // pFrame->mvuRight[i] >= 0 where the data model has stereo coordinates (the reference's Frame does; monocular mocks may not)
    template <class F> static auto has_right_impl(const F &f, int i, int) -> decltype(f.mvuRight[i] >= 0, bool()) { return !f.mvuRight.empty() && f.mvuRight[i] >= 0; }
    template <class F> static bool has_right_impl(const F &, int, long) { return false; }
    template <class F> static bool has_right_coordinate(const F &f, int i) { return has_right_impl(f, i, 0); }

public:
    static RumiOptimizer *&arena_slot() { thread_local RumiOptimizer *o = nullptr; return o; }
    static int &arena_scale() { thread_local int s = 1; return s; }
    static RumiOptimizer *arena() {
        RumiOptimizer *&o = arena_slot();
        if (!o) {
            const int k = arena_scale();
            const int rc = rumi_opt_create((1 << 16) * k, 64 * k, RUMI_OPT_MAX_KF * k, RUMI_OPT_MAX_MP * k, RUMI_OPT_MAX_EDGES * k, -1, &o);
            if (rc != RUMI_OK) { rumi_facade::report("Optimizer: optimiser arena", rc); o = nullptr; }
        }
        return o;
    }
    static bool grow_arena() {                      // a call that exceeds the arena re-creates it twice as large (rumi_status.h)
        if (arena_scale() >= 16) return false;
        if (arena_slot()) { rumi_opt_destroy(arena_slot()); arena_slot() = nullptr; }
        arena_scale() *= 2;
        return arena() != nullptr;
    }

    // int static PoseOptimization(Frame *pFrame)          Optimizer.cc:723-1001 (mono branch)
    template <class FrameT> static int PoseOptimization(FrameT *pFrame) {
        using MapPointT = typename std::remove_pointer<typename std::decay<decltype(pFrame->mvpMapPoints[0])>::type>::type;
        const int N = pFrame->N;
        std::vector<float> Xw, obs, w;
        std::vector<int> index;
        int nStereo = 0;
        {
            std::unique_lock<std::mutex> lock(MapPointT::mGlobalMutex);
            for (int i = 0; i < N; i++) {
                MapPointT *pMP = pFrame->mvpMapPoints[i];
                if (!pMP) continue;
                if (has_right_coordinate(*pFrame, i)) { nStereo++; continue; }   // EdgeStereoSE3ProjectXYZOnlyPose (:801-841): not built; reported below
                pFrame->mvbOutlier[i] = false;
                const auto &kpUn = pFrame->mvKeysUn[i];
                const auto P = pMP->GetWorldPos();
                Xw.push_back(P(0)); Xw.push_back(P(1)); Xw.push_back(P(2));
                obs.push_back(kpUn.pt.x); obs.push_back(kpUn.pt.y);
                w.push_back(pFrame->mvInvLevelSigma2[kpUn.octave]);
                index.push_back(i);
            }
        }
        if (nStereo > 0)
            rumi_facade::report("Optimizer::PoseOptimization", RUMI_E_INVALID, "observations with a right-image coordinate (mvuRight >= 0) reached the monocular PoseOptimization and were left out: the stereo edge is not built (DESIGN.md section 7)");
        const int n = (int)index.size();
        if (n < 3) return 0;
        const auto Tcw = pFrame->GetPose();
        const auto q = Tcw.unit_quaternion();
        const auto t = Tcw.translation();
        float T7[7] = {q.x(), q.y(), q.z(), q.w(), t(0), t(1), t(2)};
        const float K4[4] = {pFrame->fx, pFrame->fy, pFrame->cx, pFrame->cy};
        std::vector<uint8_t> outlier(n);
        int32_t nGood = 0;
        if (RUMI_GUARDED("Optimizer / rumi_pose_optimization", &Optimizer::grow_arena, rumi_pose_optimization(arena(), Xw.data(), obs.data(), w.data(), n, K4, T7, outlier.data(), &nGood)) != RUMI_OK) return 0;
        for (int k = 0; k < n; k++) pFrame->mvbOutlier[index[k]] = outlier[k] != 0;
#ifdef RUMI_HAVE_SOPHUS
        pFrame->SetPose(Sophus::SE3f(Eigen::Quaternionf(T7[3], T7[0], T7[1], T7[2]), Eigen::Vector3f(T7[4], T7[5], T7[6])));   // :996-998
#else
        pFrame->SetPoseFromQuatTrans(T7);      // mock data model of tests/cpp (no Eigen / Sophus in this image)
#endif
        return nGood;
    }

    // void static LocalBundleAdjustment(KeyFrame *pMainKF, vector<KeyFrame *> vpAdjustKF, vector<KeyFrame *> vpFixedKF, bool *pbStopFlag)
    // — merge / welding-window BA, Optimizer.cc:3768-4183 (monocular observations).  Needs the members the reference uses there:
    // mnBALocalForMerge on KeyFrame and MapPoint, KeyFrame::GetMapPoints(), GetMapPoint(idx).
    template <class KeyFrameT>
    static void LocalBundleAdjustment(KeyFrameT *pMainKF, std::vector<KeyFrameT *> vpAdjustKF, std::vector<KeyFrameT *> vpFixedKF, bool *pbStopFlag) {
        using MapPointT = typename std::remove_pointer<typename std::decay<decltype(pMainKF->GetMapPointMatches()[0])>::type>::type;
        auto *pCurrentMap = pMainKF->GetMap();
        std::vector<MapPointT *> vpMPs;
        std::vector<KeyFrameT *> kfs;
        std::unordered_map<KeyFrameT *, int> kfId;
        std::vector<float> kfPose;
        std::vector<uint8_t> kfFixed;
        unsigned long maxKFid = 0;
        auto add_kf = [&](KeyFrameT *k, bool fixed) {                                         // :3793-3860
            if (k->isBad() || k->GetMap() != pCurrentMap) return;
            k->mnBALocalForMerge = pMainKF->mnId;
            kfId[k] = (int)kfs.size(); kfs.push_back(k);
            const auto T = k->GetPose(); const auto q = T.unit_quaternion(); const auto t = T.translation();
            const float p[7] = {q.x(), q.y(), q.z(), q.w(), t(0), t(1), t(2)};
            kfPose.insert(kfPose.end(), p, p + 7);
            kfFixed.push_back(fixed);
            if ((unsigned long)k->mnId > maxKFid) maxKFid = (unsigned long)k->mnId;
            for (MapPointT *pMPi : k->GetMapPoints())
                if (pMPi && !pMPi->isBad() && pMPi->GetMap() == pCurrentMap && pMPi->mnBALocalForMerge != pMainKF->mnId) {
                    vpMPs.push_back(pMPi);
                    pMPi->mnBALocalForMerge = pMainKF->mnId;
                }
        };
        for (KeyFrameT *k : vpFixedKF) add_kf(k, true);
        for (KeyFrameT *k : vpAdjustKF) add_kf(k, false);
        // points that turned bad in between are skipped with their edges (:3877-3879); the rest in vpMPs order
        std::vector<MapPointT *> mps;
        std::vector<float> mpPos, eObs, eW;
        std::vector<int32_t> eMp, eKf;
        std::vector<KeyFrameT *> edgeKF;
        for (MapPointT *pMPi : vpMPs) {
            if (pMPi->isBad()) continue;
            const int p = (int)mps.size();
            mps.push_back(pMPi);
            const auto P = pMPi->GetWorldPos();
            mpPos.push_back(P(0)); mpPos.push_back(P(1)); mpPos.push_back(P(2));
            for (const auto &ob : pMPi->GetObservations()) {                                  // :3891-3925
                KeyFrameT *pKF = ob.first;
                const int idx = std::get<0>(ob.second);
                if (pKF->isBad() || (unsigned long)pKF->mnId > maxKFid || pKF->mnBALocalForMerge != pMainKF->mnId || !pKF->GetMapPoint(idx)) continue;
                if (!(pKF->mvuRight[idx] < 0)) continue;                                       // monocular observations only
                auto it = kfId.find(pKF);
                if (it == kfId.end()) continue;
                const auto &kpUn = pKF->mvKeysUn[idx];
                eMp.push_back(p); eKf.push_back(it->second);
                eObs.push_back(kpUn.pt.x); eObs.push_back(kpUn.pt.y);
                eW.push_back(pKF->mvInvLevelSigma2[kpUn.octave]);
                edgeKF.push_back(pKF);
            }
        }
        if (pbStopFlag && *pbStopFlag) return;                                                // :3982-3984
        if (kfs.empty()) return;
        const float K4[4] = {pMainKF->fx, pMainKF->fy, pMainKF->cx, pMainKF->cy};
        std::vector<uint8_t> erase(eMp.size() + 1);
        int32_t stats[4];
        if (RUMI_GUARDED("Optimizer / rumi_merge_ba", &Optimizer::grow_arena, rumi_merge_ba(arena(), (int)kfs.size(), kfPose.data(), kfFixed.data(), (int)mps.size(), mpPos.data(), (int)eMp.size(), eMp.data(),
                          eKf.data(), eObs.data(), eW.data(), K4, reinterpret_cast<const volatile uint8_t *>(pbStopFlag), erase.data(), stats)) != RUMI_OK) {
            return;
        }
        std::unique_lock<std::mutex> lock(pMainKF->GetMap()->mMutexMapUpdate);                // :4076
        for (size_t e = 0; e < eMp.size(); e++) {                                             // :4042-4085
            MapPointT *pMP = mps[eMp[e]];
            if (pMP->isBad() || !erase[e]) continue;
            edgeKF[e]->EraseMapPointMatch(pMP);
            pMP->EraseObservation(edgeKF[e]);
        }
        for (size_t k = 0; k < kfs.size(); k++) {                                             // :4108-4167: vpAdjustKF only
            if (kfFixed[k] || kfs[k]->isBad()) continue;
            const float *T7 = &kfPose[k * 7];
#ifdef RUMI_HAVE_SOPHUS
            kfs[k]->SetPose(Sophus::SE3f(Eigen::Quaternionf(T7[3], T7[0], T7[1], T7[2]), Eigen::Vector3f(T7[4], T7[5], T7[6])));
#else
            kfs[k]->SetPoseFromQuatTrans(T7);
#endif
        }
        for (size_t p = 0; p < mps.size(); p++) {                                             // :4169-4177
            if (mps[p]->isBad()) continue;
#ifdef RUMI_HAVE_SOPHUS
            mps[p]->SetWorldPos(Eigen::Vector3f(mpPos[3 * p], mpPos[3 * p + 1], mpPos[3 * p + 2]));
#else
            mps[p]->SetWorldPosXYZ(mpPos[3 * p], mpPos[3 * p + 1], mpPos[3 * p + 2]);
#endif
            mps[p]->UpdateNormalAndDepth();
        }
    }

    // void static LocalBundleAdjustment(KeyFrame *pKF, bool *pbStopFlag, Map *pMap, int &num_fixedKF, int &num_OptKF, int &num_MPs, int &num_edges)
    template <class KeyFrameT, class MapT>
    static void LocalBundleAdjustment(KeyFrameT *pKF, bool *pbStopFlag, MapT *pMap, int &num_fixedKF, int &num_OptKF, int &num_MPs, int &num_edges) {
        using MapPointT = typename std::remove_pointer<typename std::decay<decltype(pKF->GetMapPointMatches()[0])>::type>::type;
        // Local KeyFrames: first breadth search from the current key-frame (:1004-1017)
        std::list<KeyFrameT *> lLocalKeyFrames;
        lLocalKeyFrames.push_back(pKF);
        pKF->mnBALocalForKF = pKF->mnId;
        auto *pCurrentMap = pKF->GetMap();
        for (KeyFrameT *pKFi : pKF->GetVectorCovisibleKeyFrames()) {
            pKFi->mnBALocalForKF = pKF->mnId;
            if (!pKFi->isBad() && pKFi->GetMap() == pCurrentMap) lLocalKeyFrames.push_back(pKFi);
        }
        // Local MapPoints seen in local key-frames (:1019-1039)
        num_fixedKF = 0;
        std::list<MapPointT *> lLocalMapPoints;
        for (KeyFrameT *pKFi : lLocalKeyFrames) {
            if (pKFi->mnId == pMap->GetInitKFid()) num_fixedKF = 1;
            for (MapPointT *pMP : pKFi->GetMapPointMatches())
                if (pMP && !pMP->isBad() && pMP->GetMap() == pCurrentMap && pMP->mnBALocalForKF != pKF->mnId) {
                    lLocalMapPoints.push_back(pMP);
                    pMP->mnBALocalForKF = pKF->mnId;
                }
        }
        // Fixed key-frames: see local map points but are not local (:1041-1055)
        std::list<KeyFrameT *> lFixedCameras;
        for (MapPointT *pMP : lLocalMapPoints)
            for (const auto &ob : pMP->GetObservations()) {
                KeyFrameT *pKFi = ob.first;
                if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) {
                    pKFi->mnBAFixedForKF = pKF->mnId;
                    if (!pKFi->isBad() && pKFi->GetMap() == pCurrentMap) lFixedCameras.push_back(pKFi);
                }
            }
        num_fixedKF = (int)lFixedCameras.size() + num_fixedKF;
        if (num_fixedKF == 0) return;                                                        // :1057-1060

        // flatten: vertices then edges in the reference's construction order (:1088-1271)
        std::vector<KeyFrameT *> kfs;
        std::unordered_map<KeyFrameT *, int> kfId;
        std::vector<float> kfPose;
        std::vector<uint8_t> kfFixed;
        auto add_kf = [&](KeyFrameT *k, bool fixed) {
            kfId[k] = (int)kfs.size(); kfs.push_back(k);
            const auto T = k->GetPose(); const auto q = T.unit_quaternion(); const auto t = T.translation();
            const float p[7] = {q.x(), q.y(), q.z(), q.w(), t(0), t(1), t(2)};
            kfPose.insert(kfPose.end(), p, p + 7);
            kfFixed.push_back(fixed);
        };
        for (KeyFrameT *k : lLocalKeyFrames) add_kf(k, k->mnId == pMap->GetInitKFid());
        num_OptKF = (int)lLocalKeyFrames.size();
        for (KeyFrameT *k : lFixedCameras) add_kf(k, true);
        std::vector<MapPointT *> mps(lLocalMapPoints.begin(), lLocalMapPoints.end());
        std::vector<float> mpPos, eObs, eW;
        std::vector<int32_t> eMp, eKf;
        std::vector<KeyFrameT *> edgeKF;
        for (size_t p = 0; p < mps.size(); p++) {
            const auto P = mps[p]->GetWorldPos();
            mpPos.push_back(P(0)); mpPos.push_back(P(1)); mpPos.push_back(P(2));
            for (const auto &ob : mps[p]->GetObservations()) {
                KeyFrameT *pKFi = ob.first;
                if (pKFi->isBad() || pKFi->GetMap() != pCurrentMap) continue;
                const int leftIndex = std::get<0>(ob.second);
                if (leftIndex == -1 || !(pKFi->mvuRight[leftIndex] < 0)) continue;             // monocular observations only
                auto it = kfId.find(pKFi);
                if (it == kfId.end()) continue;
                const auto &kpUn = pKFi->mvKeysUn[leftIndex];
                eMp.push_back((int32_t)p); eKf.push_back(it->second);
                eObs.push_back(kpUn.pt.x); eObs.push_back(kpUn.pt.y);
                eW.push_back(pKFi->mvInvLevelSigma2[kpUn.octave]);
                edgeKF.push_back(pKFi);
            }
        }
        num_MPs = (int)mps.size();
        num_edges = (int)eMp.size();
        if (pbStopFlag && *pbStopFlag) return;                                                // :1274-1276
        const float K4[4] = {pKF->fx, pKF->fy, pKF->cx, pKF->cy};
        std::vector<uint8_t> erase(eMp.size() + 1);
        int32_t stats[4];
        if (RUMI_GUARDED("Optimizer / rumi_local_ba", &Optimizer::grow_arena, rumi_local_ba(arena(), (int)kfs.size(), kfPose.data(), kfFixed.data(), (int)mps.size(), mpPos.data(), (int)eMp.size(), eMp.data(),
                          eKf.data(), eObs.data(), eW.data(), K4, reinterpret_cast<const volatile uint8_t *>(pbStopFlag), erase.data(), stats)) != RUMI_OK)
            return;
        if (stats[3]) return;
        std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);                             // :1325
        for (size_t e = 0; e < eMp.size(); e++) {
            MapPointT *pMP = mps[eMp[e]];
            if (pMP->isBad() || !erase[e]) continue;
            edgeKF[e]->EraseMapPointMatch(pMP);
            pMP->EraseObservation(edgeKF[e]);
        }
        for (size_t k = 0; k < kfs.size(); k++)
            if ((int)k < num_OptKF) {                                                        // :1338-1344
                const float *T7 = &kfPose[k * 7];
#ifdef RUMI_HAVE_SOPHUS
                kfs[k]->SetPose(Sophus::SE3f(Eigen::Quaternionf(T7[3], T7[0], T7[1], T7[2]), Eigen::Vector3f(T7[4], T7[5], T7[6])));
#else
                kfs[k]->SetPoseFromQuatTrans(T7);
#endif
            }
        for (size_t p = 0; p < mps.size(); p++) {
#ifdef RUMI_HAVE_SOPHUS
            mps[p]->SetWorldPos(Eigen::Vector3f(mpPos[3 * p], mpPos[3 * p + 1], mpPos[3 * p + 2]));                              // :1347-1351
#else
            mps[p]->SetWorldPosXYZ(mpPos[3 * p], mpPos[3 * p + 1], mpPos[3 * p + 2]);
#endif
            mps[p]->UpdateNormalAndDepth();
        }
        pMap->IncreaseChangeIndex();
    }

#ifdef RUMI_HAVE_SOPHUS
    // void static BundleAdjustment(const vector<KeyFrame *> &vpKFs, const vector<MapPoint *> &vpMP, int nIterations = 5, bool *pbStopFlag = NULL,
    //                              const unsigned long nLoopKF = 0, const bool bRobust = true)                Optimizer.cc:54-351 (monocular edges)
    template <class KeyFrameT, class MapPointT>
    static void BundleAdjustment(const std::vector<KeyFrameT *> &vpKFs, const std::vector<MapPointT *> &vpMP, int nIterations = 5, bool *pbStopFlag = nullptr,
                                 const unsigned long nLoopKF = 0, const bool bRobust = true) {
        std::vector<bool> vbNotIncludedMP(vpMP.size(), false);
        auto *pMap = vpKFs[0]->GetMap();
        std::vector<KeyFrameT *> kfs;
        std::unordered_map<KeyFrameT *, int> kfId;
        std::vector<float> kfPose;
        std::vector<uint8_t> kfFixed;
        unsigned long maxKFid = 0;
        for (KeyFrameT *pKF : vpKFs) {                                                        // :103-115
            if (pKF->isBad()) continue;
            kfId[pKF] = (int)kfs.size(); kfs.push_back(pKF);
            const auto T = pKF->GetPose(); const auto q = T.unit_quaternion(); const auto t = T.translation();
            const float p7[7] = {q.x(), q.y(), q.z(), q.w(), t(0), t(1), t(2)};
            kfPose.insert(kfPose.end(), p7, p7 + 7);
            kfFixed.push_back(pKF->mnId == pMap->GetInitKFid());
            if ((unsigned long)pKF->mnId > maxKFid) maxKFid = pKF->mnId;
        }
        std::vector<MapPointT *> mps;
        std::vector<size_t> mpIndex;
        std::vector<float> mpPos, eObs, eW;
        std::vector<int32_t> eMp, eKf;
        for (size_t i = 0; i < vpMP.size(); i++) {                                            // :123-253
            MapPointT *pMP = vpMP[i];
            if (pMP->isBad()) continue;
            const auto P = pMP->GetWorldPos();
            int nEdges = 0;
            for (const auto &ob : pMP->GetObservations()) {
                KeyFrameT *pKF = ob.first;
                if (pKF->isBad() || (unsigned long)pKF->mnId > maxKFid) continue;
                auto it = kfId.find(pKF);
                if (it == kfId.end()) continue;                                               // optimizer.vertex(pKF->mnId) == NULL
                nEdges++;
                const int leftIndex = std::get<0>(ob.second);
                if (leftIndex == -1 || !(pKF->mvuRight[leftIndex] < 0)) continue;             // monocular observations only
                const auto &kpUn = pKF->mvKeysUn[leftIndex];
                eMp.push_back((int32_t)mps.size()); eKf.push_back(it->second);
                eObs.push_back(kpUn.pt.x); eObs.push_back(kpUn.pt.y);
                eW.push_back(pKF->mvInvLevelSigma2[kpUn.octave]);
            }
            if (nEdges == 0) { vbNotIncludedMP[i] = true; continue; }                         // :251-254: the vertex is removed again
            mps.push_back(pMP); mpIndex.push_back(i);
            mpPos.push_back(P(0)); mpPos.push_back(P(1)); mpPos.push_back(P(2));
        }
        if (kfs.empty()) return;
        const float K4[4] = {kfs[0]->fx, kfs[0]->fy, kfs[0]->cx, kfs[0]->cy};
        int32_t stats[4];
        if (RUMI_GUARDED("Optimizer / rumi_bundle_adjustment", &Optimizer::grow_arena, rumi_bundle_adjustment(arena(), (int)kfs.size(), kfPose.data(), kfFixed.data(), (int)mps.size(), mpPos.data(), (int)eMp.size(), eMp.data(), eKf.data(),
                                   eObs.data(), eW.data(), K4, reinterpret_cast<const volatile uint8_t *>(pbStopFlag), nIterations, bRobust, stats)) != RUMI_OK) {
            return;
        }
        const bool direct = nLoopKF == (unsigned long)pMap->GetOriginKF()->mnId;
        for (size_t k = 0; k < kfs.size(); k++) {                                             // :264-327
            const float *T7 = &kfPose[k * 7];
            const Sophus::SE3f T(Eigen::Quaternionf(T7[3], T7[0], T7[1], T7[2]), Eigen::Vector3f(T7[4], T7[5], T7[6]));
            if (direct) kfs[k]->SetPose(T);
            else { kfs[k]->mTcwGBA = T; kfs[k]->mnBAGlobalForKF = nLoopKF; }
        }
        for (size_t p = 0; p < mps.size(); p++) {                                             // :330-349
            const Eigen::Vector3f X(mpPos[3 * p], mpPos[3 * p + 1], mpPos[3 * p + 2]);
            if (direct) { mps[p]->SetWorldPos(X); mps[p]->UpdateNormalAndDepth(); }
            else { mps[p]->mPosGBA = X; mps[p]->mnBAGlobalForKF = nLoopKF; }
        }
    }

    // void static GlobalBundleAdjustemnt(Map *pMap, int nIterations = 5, bool *pbStopFlag = NULL, const unsigned long nLoopKF = 0, const bool bRobust = true)   :48-52
    template <class MapT>
    static void GlobalBundleAdjustemnt(MapT *pMap, int nIterations = 5, bool *pbStopFlag = nullptr, const unsigned long nLoopKF = 0, const bool bRobust = true) {
        BundleAdjustment(pMap->GetAllKeyFrames(), pMap->GetAllMapPoints(), nIterations, pbStopFlag, nLoopKF, bRobust);
    }
#endif

#ifdef RUMI_HAVE_SOPHUS
    // Gathers one correspondence of OptimizeSim3 / OptimizeCloudSim3 (Optimizer.cc:1972-2100, :2246-2392); false when the reference skips it.
    struct Sim3Gather {
        std::vector<float> P1c, P2c, obs1, obs2, w1, w2;
        std::vector<uint8_t> skip12, skip21;
        std::vector<int32_t> pairOf;
        std::vector<size_t> index;             // vnIndexEdge
        int n() const { return (int)w1.size(); }
    };
    template <class KeyFrameT, class MapPointT>
    static bool sim3_gather(Sim3Gather &g, KeyFrameT *pKF1, KeyFrameT *pKF2, MapPointT *pMP1, MapPointT *pMP2, size_t i, int pair, bool bAllPoints) {
        if (!pMP1 || !pMP2) return false;                                              // "match without map point": vertices only, no edges (:2016-2031)
        if (pMP1->isBad() || pMP2->isBad()) return false;                              // :2011-2014
        const int i2 = std::get<0>(pMP2->GetIndexInKeyFrame(pKF2));
        const auto P3D1c = pKF1->GetRotation() * pMP1->GetWorldPos() + pKF1->GetTranslation();   // :1990, :1997
        const auto P3D2c = pKF2->GetRotation() * pMP2->GetWorldPos() + pKF2->GetTranslation();
        if (i2 < 0 && !bAllPoints) return false;                                       // :2033-2036
        if (P3D2c(2) < 0) return false;                                                // :2038-2041
        const auto &kpUn1 = pKF1->mvKeysUn[i];
        float ox, oy; int octave2;
        if (i2 >= 0) { const auto &kpUn2 = pKF2->mvKeysUn[i2]; ox = kpUn2.pt.x; oy = kpUn2.pt.y; octave2 = kpUn2.octave; }
        else { const float invz = 1 / P3D2c(2); ox = P3D2c(0) * invz; oy = P3D2c(1) * invz; octave2 = pMP2->mnTrackScaleLevel; }   // :2065-2071
        for (int c = 0; c < 3; c++) { g.P1c.push_back(P3D1c(c)); g.P2c.push_back(P3D2c(c)); }
        g.obs1.push_back(kpUn1.pt.x); g.obs1.push_back(kpUn1.pt.y); g.obs2.push_back(ox); g.obs2.push_back(oy);
        g.w1.push_back(pKF1->mvInvLevelSigma2[kpUn1.octave]); g.w2.push_back(pKF2->mvInvLevelSigma2[octave2]);
        g.skip12.push_back(pMP2->isEdge); g.skip21.push_back(pMP1->isEdge);
        g.pairOf.push_back(pair); g.index.push_back(i);
        return true;
    }
    template <class Sim3T> static void sim3_to8(const Sim3T &S, double *o) {
        const auto q = S.rotation(); const auto t = S.translation();
        o[0] = q.x(); o[1] = q.y(); o[2] = q.z(); o[3] = q.w(); o[4] = t(0); o[5] = t(1); o[6] = t(2); o[7] = S.scale();
    }
    template <class Sim3T> static Sim3T sim3_from8(const double *S) {
        return Sim3T(Eigen::Quaterniond(S[3], S[0], S[1], S[2]), Eigen::Vector3d(S[4], S[5], S[6]), S[7]);
    }

    // static int OptimizeSim3(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint *> &vpMatches1, g2o::Sim3 &g2oS12, const float th2,
    //                         const bool bFixScale, Eigen::Matrix<double, 7, 7> &mAcumHessian, const bool bAllPoints)      Optimizer.cc:1920-2167
    template <class KeyFrameT, class MapPointT, class Sim3T, class HessianT>
    static int OptimizeSim3(KeyFrameT *pKF1, KeyFrameT *pKF2, std::vector<MapPointT *> &vpMatches1, Sim3T &g2oS12, const float th2, const bool bFixScale,
                            HessianT &mAcumHessian, const bool bAllPoints = false) {
        const int N = (int)vpMatches1.size();
        const auto vpMapPoints1 = pKF1->GetMapPointMatches();
        Sim3Gather g;
        for (int i = 0; i < N; i++) {
            if (!vpMatches1[i]) continue;
            sim3_gather(g, pKF1, pKF2, vpMapPoints1[i], vpMatches1[i], (size_t)i, 0, bAllPoints);
        }
        // the isEdge tests belong to the cloud variant only: here both edges always exist
        const float K1[4] = {pKF1->fx, pKF1->fy, pKF1->cx, pKF1->cy}, K2[4] = {pKF2->fx, pKF2->fy, pKF2->cx, pKF2->cy};
        double S[8];
        sim3_to8(g2oS12, S);
        std::vector<uint8_t> status(g.n() + 1);
        int32_t res[3] = {0, 0, 1};
        if (RUMI_GUARDED("Optimizer / rumi_optimize_sim3", &Optimizer::grow_arena, rumi_optimize_sim3(arena(), g.n(), nullptr, 0, nullptr, nullptr, g.P1c.data(), g.P2c.data(), g.obs1.data(), g.obs2.data(), g.w1.data(), g.w2.data(),
                               nullptr, nullptr, K1, K2, th2, bFixScale, 1, S, status.data(), res)) != RUMI_OK) {
            std::fprintf(stderr, "OptimizeSim3: %s\n", rumi_last_error());
            return 0;
        }
        for (int k = 0; k < g.n(); k++)
            if (status[k] == 1 || (!res[2] && status[k] == 2)) vpMatches1[g.index[k]] = static_cast<MapPointT *>(nullptr);   // :2112, :2156
        if (res[2]) return 0;                                                          // :2135-2136: g2oS12 keeps its value
        mAcumHessian.setZero();                                                        // :2144 (never accumulated upstream)
        g2oS12 = sim3_from8<Sim3T>(S);                                                 // :2163-2164
        return res[0];
    }

    // static float OptimizeCloudSim3(const vector<KeyFrame *> &map1KFs, const vector<KeyFrame *> &map2KFs, const vector<vector<MapPoint *>> &avpMatches,
    //                                g2o::Sim3 &gSw1w2, const float th2, const bool bFixScale, Eigen::Matrix<double, 7, 7> &mAcumHessian,
    //                                const bool bAllPoints)                                                              Optimizer.cc:2169-2471
    template <class KeyFrameT, class MapPointT, class Sim3T, class HessianT>
    static float OptimizeCloudSim3(const std::vector<KeyFrameT *> &map1KFs, const std::vector<KeyFrameT *> &map2KFs,
                                   const std::vector<std::vector<MapPointT *>> &avpMatches, Sim3T &gSw1w2, const float th2, const bool bFixScale,
                                   HessianT &mAcumHessian, const bool bAllPoints = false) {
        Sim3Gather g;
        std::vector<double> A, B;
        for (size_t k = 0; k < map2KFs.size(); ++k) {
            KeyFrameT *pKF1 = map1KFs[k], *pKF2 = map2KFs[k];
            const Sim3T gSc1w(pKF1->GetRotation().template cast<double>(), pKF1->GetTranslation().template cast<double>(), 1.0);   // :2231-2232
            const Sim3T gSc2w(pKF2->GetRotation().template cast<double>(), pKF2->GetTranslation().template cast<double>(), 1.0);
            double a[8], b[8];
            sim3_to8(gSc1w, a); sim3_to8(gSc2w, b);
            A.insert(A.end(), a, a + 8); B.insert(B.end(), b, b + 8);
            const auto &vpMatches1 = avpMatches[k];
            const auto vpMapPoints1 = pKF1->GetMapPointMatches();
            for (size_t i = 0; i < vpMatches1.size(); i++) {
                if (!vpMapPoints1[i] || !vpMatches1[i]) continue;                       // :2251-2252
                sim3_gather(g, pKF1, pKF2, vpMapPoints1[i], vpMatches1[i], i, (int)k, bAllPoints);
            }
        }
        if (map2KFs.empty()) return 0.f;
        const float K1[4] = {map1KFs[0]->fx, map1KFs[0]->fy, map1KFs[0]->cx, map1KFs[0]->cy};   // vSim3->pCamera1 = map1KFs[0]->mpCamera (:2192-2193)
        const float K2[4] = {map2KFs[0]->fx, map2KFs[0]->fy, map2KFs[0]->cx, map2KFs[0]->cy};
        double S[8];
        sim3_to8(gSw1w2, S);
        std::vector<uint8_t> status(g.n() + 1);
        int32_t res[3] = {0, 0, 1};
        if (RUMI_GUARDED("Optimizer / rumi_optimize_sim3", &Optimizer::grow_arena, rumi_optimize_sim3(arena(), g.n(), g.pairOf.data(), (int32_t)map2KFs.size(), A.data(), B.data(), g.P1c.data(), g.P2c.data(), g.obs1.data(),
                               g.obs2.data(), g.w1.data(), g.w2.data(), g.skip12.data(), g.skip21.data(), K1, K2, th2, bFixScale, 0, S, status.data(), res)) != RUMI_OK) {
            std::fprintf(stderr, "OptimizeCloudSim3: %s\n", rumi_last_error());
            return 0.f;
        }
        gSw1w2 = sim3_from8<Sim3T>(S);                                                 // :2397 (before the early return) and :2466
        if (res[2]) return 0;                                                          // :2437-2438
        mAcumHessian.setZero();                                                        // :2446
        return (float)res[0] / (float)g.n();                                           // :2468
    }
#endif
};

}  // namespace RUMI_FACADE_NAMESPACE
