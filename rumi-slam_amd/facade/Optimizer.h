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

namespace ORB_SLAM3 {

class Optimizer {
public:
    static RumiOptimizer *arena() {
        thread_local RumiOptimizer *o = nullptr;
        if (!o && rumi_opt_create(1 << 16, 64, 256, 1 << 16, 1 << 20, -1, &o) != RUMI_OK) return nullptr;
        return o;
    }

    // int static PoseOptimization(Frame *pFrame)          Optimizer.cc:723-1001 (mono branch)
    template <class FrameT> static int PoseOptimization(FrameT *pFrame) {
        using MapPointT = typename std::remove_pointer<typename std::decay<decltype(pFrame->mvpMapPoints[0])>::type>::type;
        const int N = pFrame->N;
        std::vector<float> Xw, obs, w;
        std::vector<int> index;
        {
            std::unique_lock<std::mutex> lock(MapPointT::mGlobalMutex);
            for (int i = 0; i < N; i++) {
                MapPointT *pMP = pFrame->mvpMapPoints[i];
                if (!pMP) continue;
                pFrame->mvbOutlier[i] = false;
                const auto &kpUn = pFrame->mvKeysUn[i];
                const auto P = pMP->GetWorldPos();
                Xw.push_back(P(0)); Xw.push_back(P(1)); Xw.push_back(P(2));
                obs.push_back(kpUn.pt.x); obs.push_back(kpUn.pt.y);
                w.push_back(pFrame->mvInvLevelSigma2[kpUn.octave]);
                index.push_back(i);
            }
        }
        const int n = (int)index.size();
        if (n < 3) return 0;
        const auto Tcw = pFrame->GetPose();
        const auto q = Tcw.unit_quaternion();
        const auto t = Tcw.translation();
        float T7[7] = {q.x(), q.y(), q.z(), q.w(), t(0), t(1), t(2)};
        const float K4[4] = {pFrame->fx, pFrame->fy, pFrame->cx, pFrame->cy};
        std::vector<uint8_t> outlier(n);
        int32_t nGood = 0;
        if (rumi_pose_optimization(arena(), Xw.data(), obs.data(), w.data(), n, K4, T7, outlier.data(), &nGood) != RUMI_OK) return 0;
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
        if (rumi_merge_ba(arena(), (int)kfs.size(), kfPose.data(), kfFixed.data(), (int)mps.size(), mpPos.data(), (int)eMp.size(), eMp.data(),
                          eKf.data(), eObs.data(), eW.data(), K4, reinterpret_cast<const volatile uint8_t *>(pbStopFlag), erase.data(), stats) != RUMI_OK) {
            std::fprintf(stderr, "Optimizer::LocalBundleAdjustment (merge): %s\n", rumi_last_error());
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
        if (rumi_local_ba(arena(), (int)kfs.size(), kfPose.data(), kfFixed.data(), (int)mps.size(), mpPos.data(), (int)eMp.size(), eMp.data(),
                          eKf.data(), eObs.data(), eW.data(), K4, reinterpret_cast<const volatile uint8_t *>(pbStopFlag), erase.data(), stats) != RUMI_OK)
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
};

}  // namespace ORB_SLAM3
