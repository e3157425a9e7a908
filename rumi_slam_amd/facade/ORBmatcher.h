// Drop-in ORB_SLAM3::ORBmatcher over the MI355X C ABI (include/rumi_match.h).
// Class surface: R/include/cloud_edge_slam_lib/ORBmatcher.h:36-103.  The three hot searches are member templates over the
// data-model types so that this header compiles both against the reference's Frame / KeyFrame / MapPoint (same member
// names are used: mvKeysUn, mDescriptors, mvpMapPoints, mbTrackInView, ...) and against the small mock types of
// tests/cpp.  Everything here is marshalling: pointers become indices, results are written back into the caller's vectors.
#pragma once
#include <map>
#include <set>
#include <unordered_map>
#include <vector>

#include "cv_shim.h"
#include "rumi_match.h"
#include "rumi_status.h"

// The classes live in ORB_SLAM3 when these headers REPLACE the reference's (member templates deduce the reference's own types at every call
// site), or in a namespace of their own when the reference's headers stay and facade/shells/*.cc forward to them (-DRUMI_FACADE_NAMESPACE=...).
#ifndef RUMI_FACADE_NAMESPACE
#define RUMI_FACADE_NAMESPACE ORB_SLAM3
#endif
namespace RUMI_FACADE_NAMESPACE {

class ORBmatcher {
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

    static const int TH_LOW = RUMI_TH_LOW;
    static const int TH_HIGH = RUMI_TH_HIGH;
    static const int HISTO_LENGTH = RUMI_HISTO_LENGTH;

    // Computes the Hamming distance between two ORB descriptors
    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b) { return rumi_descriptor_distance(a.ptr(0), b.ptr(0)); }

    // One matcher arena per calling thread (handles are not re-entrant).
    // (8192 features / 32768 queries to begin with; a call that needs more re-creates it twice as large -- features up to the library's
    // 16384 per frame, queries up to 16 x: rumi_status.h)
    static RumiMatcher *&arena_slot() { thread_local RumiMatcher *m = nullptr; return m; }
    static int &arena_scale() { thread_local int s = 1; return s; }
    static RumiMatcher *arena() {
        RumiMatcher *&m = arena_slot();
        if (!m) {
            const int rc = rumi_match_create(8192 * (arena_scale() > 1 ? 2 : 1), 32768 * arena_scale(), -1, &m);
            if (rc != RUMI_OK) { rumi_facade::report("ORBmatcher: matcher arena", rc); m = nullptr; }
        }
        return m;
    }
    static bool grow_arena() {
        if (arena_scale() >= 16) return false;
        if (arena_slot()) { rumi_match_destroy(arena_slot()); arena_slot() = nullptr; }
        arena_scale() *= 2;
        return arena() != nullptr;
    }

    // Search matches between Frame keypoints and projected MapPoints. Returns number of matches (TrackLocalMap)
    template <class FrameT, class MapPointT>
    int SearchByProjection(FrameT &F, const std::vector<MapPointT *> &vpMapPoints, const float th = 3, const bool bFarPoints = false,
                           const float thFarPoints = 50.0f) {
        const int nmp = (int)vpMapPoints.size();
        std::vector<uint8_t> inView(nmp), bad(nmp), desc((size_t)nmp * 32);
        std::vector<float> px(nmp), py(nmp), vc(nmp), depth(nmp);
        std::vector<int32_t> lvl(nmp), obs(nmp);
        std::unordered_map<const MapPointT *, int> idOf;
        std::vector<MapPointT *> byId(vpMapPoints);
        for (int i = 0; i < nmp; i++) {
            MapPointT *p = vpMapPoints[i];
            idOf.emplace(p, i);
            inView[i] = p->mbTrackInView; bad[i] = p->isBad();
            px[i] = p->mTrackProjX; py[i] = p->mTrackProjY; vc[i] = p->mTrackViewCos; depth[i] = p->mTrackDepth;
            lvl[i] = p->mnTrackScaleLevel; obs[i] = p->Observations();
            const cv::Mat d = p->GetDescriptor();
            std::memcpy(&desc[(size_t)i * 32], d.ptr(0), 32);
        }
        // features that already hold a map point: ids beyond nmp for points that are not in vpMapPoints
        std::vector<int32_t> frameMp(F.N, -1);
        for (int f = 0; f < F.N; f++) {
            MapPointT *p = F.mvpMapPoints[f];
            if (!p) continue;
            auto it = idOf.find(p);
            if (it != idOf.end()) frameMp[f] = it->second;
            else { frameMp[f] = (int)byId.size(); idOf.emplace(p, (int)byId.size()); byId.push_back(p); obs.push_back(p->Observations()); }
        }
        // the extra ids only need Observations(): pad the per-point arrays the device indexes by id
        const int nid = (int)byId.size();
        std::vector<int32_t> obsAll(obs);
        RumiFrameFeatures fv = view(F);
        int32_t nmatches = 0;
        // the kernel reads mp_obs[id] for every id a feature may hold, so pass the extended array; queries are the first nmp ids
        inView.resize(nid, 0); bad.resize(nid, 1); px.resize(nid, 0); py.resize(nid, 0); vc.resize(nid, 0); depth.resize(nid, 0);
        lvl.resize(nid, 0); desc.resize((size_t)nid * 32, 0);
        if (RUMI_GUARDED("ORBmatcher / rumi_search_by_projection_mappoints", &ORBmatcher::grow_arena, rumi_search_by_projection_mappoints(arena(), &fv, nid, inView.data(), px.data(), py.data(), lvl.data(), vc.data(), depth.data(),
                                                bad.data(), desc.data(), obsAll.data(), th, bFarPoints, thFarPoints, mfNNratio,
                                                frameMp.data(), &nmatches)) != RUMI_OK)
            return -1;
        for (int f = 0; f < F.N; f++) F.mvpMapPoints[f] = frameMp[f] >= 0 ? byId[frameMp[f]] : nullptr;
        return nmatches;
    }

    // Project MapPoints tracked in last frame into the current frame and search matches (TrackWithMotionModel)
    template <class FrameT>
    int SearchByProjection(FrameT &CurrentFrame, const FrameT &LastFrame, const float th, const bool /*bMono*/) {
        using MapPointT = typename std::remove_pointer<typename std::decay<decltype(CurrentFrame.mvpMapPoints[0])>::type>::type;
        std::unordered_map<const MapPointT *, int> idOf;
        std::vector<MapPointT *> byId;
        auto id_of = [&](MapPointT *p) { auto it = idOf.find(p); if (it != idOf.end()) return it->second; idOf.emplace(p, (int)byId.size()); byId.push_back(p); return (int)byId.size() - 1; };
        std::vector<int32_t> lastMp(LastFrame.N, -1), curMp(CurrentFrame.N, -1);
        std::vector<uint8_t> lastOut(LastFrame.N, 0);
        for (int i = 0; i < LastFrame.N; i++) { if (LastFrame.mvpMapPoints[i]) lastMp[i] = id_of(LastFrame.mvpMapPoints[i]); lastOut[i] = LastFrame.mvbOutlier[i]; }
        for (int i = 0; i < CurrentFrame.N; i++) if (CurrentFrame.mvpMapPoints[i]) curMp[i] = id_of(CurrentFrame.mvpMapPoints[i]);
        const int nmp = (int)byId.size();
        std::vector<float> pos((size_t)nmp * 3);
        std::vector<uint8_t> desc((size_t)nmp * 32);
        std::vector<int32_t> obs(nmp);
        for (int i = 0; i < nmp; i++) {
            const auto P = byId[i]->GetWorldPos();
            pos[3 * i] = P(0); pos[3 * i + 1] = P(1); pos[3 * i + 2] = P(2);
            const cv::Mat d = byId[i]->GetDescriptor();
            std::memcpy(&desc[(size_t)i * 32], d.ptr(0), 32);
            obs[i] = byId[i]->Observations();
        }
        const auto Tcw = CurrentFrame.GetPose();
        const auto q = Tcw.unit_quaternion();
        const auto t = Tcw.translation();
        const float T7[7] = {q.x(), q.y(), q.z(), q.w(), t(0), t(1), t(2)};
        const float K4[4] = {CurrentFrame.fx, CurrentFrame.fy, CurrentFrame.cx, CurrentFrame.cy};
        RumiFrameFeatures cv_ = view(CurrentFrame);
        int32_t nmatches = 0;
        if (RUMI_GUARDED("ORBmatcher / rumi_search_by_projection_frame", &ORBmatcher::grow_arena, rumi_search_by_projection_frame(arena(), &cv_, T7, K4, reinterpret_cast<const RumiKeyPoint *>(LastFrame.mvKeysUn.data()), LastFrame.N,
                                            lastMp.data(), lastOut.data(), nmp, pos.data(), desc.data(), obs.data(), th, mbCheckOrientation,
                                            curMp.data(), &nmatches)) != RUMI_OK)
            return -1;
        for (int f = 0; f < CurrentFrame.N; f++) CurrentFrame.mvpMapPoints[f] = curMp[f] >= 0 ? byId[curMp[f]] : nullptr;
        return nmatches;
    }

    // Matching for the Map Initialization (only used in the monocular case) — ORBmatcher.cc:581-680
    template <class FrameT, class Point2fT>
    int SearchForInitialization(FrameT &F1, FrameT &F2, std::vector<Point2fT> &vbPrevMatched, std::vector<int> &vnMatches12, int windowSize = 10) {
        static_assert(sizeof(Point2fT) == 2 * sizeof(float), "cv::Point2f layout expected");
        RumiFrameFeatures a = view(F1), b = view(F2);
        a.n = (int32_t)F1.mvKeysUn.size(); b.n = (int32_t)F2.mvKeysUn.size();
        vnMatches12.assign(F1.mvKeysUn.size(), -1);
        int32_t nmatches = 0;
        if (RUMI_GUARDED("ORBmatcher / rumi_search_for_initialization", &ORBmatcher::grow_arena, rumi_search_for_initialization(arena(), &a, &b, reinterpret_cast<float *>(vbPrevMatched.data()), windowSize, mfNNratio, mbCheckOrientation,
                                           vnMatches12.data(), &nmatches)) != RUMI_OK)
            return -1;
        return nmatches;
    }

    // Search matches between MapPoints in a KeyFrame and ORB in a Frame, by vocabulary node (TrackReferenceKeyFrame, Relocalization)
    template <class KeyFrameT, class FrameT, class MapPointT>
    int SearchByBoW(KeyFrameT *pKF, FrameT &F, std::vector<MapPointT *> &vpMapPointMatches) {
        const std::vector<MapPointT *> vpMapPointsKF = pKF->GetMapPointMatches();
        std::vector<int32_t> kfMp(vpMapPointsKF.size(), -1);
        std::vector<uint8_t> bad(vpMapPointsKF.size(), 0);
        for (size_t i = 0; i < vpMapPointsKF.size(); i++)
            if (vpMapPointsKF[i]) { kfMp[i] = (int)i; bad[i] = vpMapPointsKF[i]->isBad(); }     // id = key-frame feature index
        Csr a = csr(pKF->mFeatVec), b = csr(F.mFeatVec);
        RumiFrameFeatures kv = view(*pKF), fv = view(F);
        std::vector<int32_t> matches(F.N, -1);
        int32_t nmatches = 0;
        if (RUMI_GUARDED("ORBmatcher / rumi_search_by_bow", &ORBmatcher::grow_arena, rumi_search_by_bow(arena(), &kv, &a.v, kfMp.data(), (int)bad.size(), bad.data(), &fv, &b.v, mfNNratio, mbCheckOrientation,
                               matches.data(), &nmatches)) != RUMI_OK)
            return -1;
        vpMapPointMatches.assign(F.N, static_cast<MapPointT *>(nullptr));
        for (int f = 0; f < F.N; f++) if (matches[f] >= 0) vpMapPointMatches[f] = vpMapPointsKF[matches[f]];
        return nmatches;
    }

    // The walk of Tracking::Relocalization over its candidate key-frames (Tracking.cc:3240-3260: one SearchByBoW(pKF, mCurrentFrame,
    // vvpMapPointMatches[i]) per candidate, each from an empty result) as ONE device call.  Candidates that are NULL or isBad() are left out
    // (their entry of the result is -1 and their match vector stays untouched, as the caller's own `if (pKF->isBad())` branch has it).
    // Returns the match count per candidate (-1: skipped or failed).
    template <class KeyFrameT, class FrameT, class MapPointT>
    std::vector<int> SearchByBoW(const std::vector<KeyFrameT *> &vpCandidateKFs, FrameT &F, std::vector<std::vector<MapPointT *>> &vvpMapPointMatches) {
        const int nKFs = (int)vpCandidateKFs.size();
        std::vector<int> result(nKFs, -1);
        vvpMapPointMatches.resize(nKFs);
        std::vector<int> live;
        std::vector<std::vector<MapPointT *>> mpsKF;
        std::vector<std::vector<int32_t>> kfMp;
        std::vector<std::vector<uint8_t>> bad;
        std::vector<Csr> fvs;
        std::vector<RumiFrameFeatures> views;
        for (int i = 0; i < nKFs; i++) {
            KeyFrameT *pKF = vpCandidateKFs[i];
            if (!pKF || pKF->isBad()) continue;
            live.push_back(i);
            mpsKF.push_back(pKF->GetMapPointMatches());
            const std::vector<MapPointT *> &v = mpsKF.back();
            kfMp.emplace_back(v.size(), -1); bad.emplace_back(v.size(), 0);
            for (size_t j = 0; j < v.size(); j++)
                if (v[j]) { kfMp.back()[j] = (int)j; bad.back()[j] = v[j]->isBad(); }
            fvs.push_back(csr(pKF->mFeatVec));
            views.push_back(view(*pKF));
        }
        const int K = (int)live.size();
        if (K == 0) return result;
        std::vector<RumiFeatureVector> fvv(K);
        std::vector<const int32_t *> mpPtr(K);
        std::vector<const uint8_t *> badPtr(K);
        std::vector<int32_t> nmp(K);
        for (int k = 0; k < K; k++) { fvv[k] = fvs[k].v; mpPtr[k] = kfMp[k].data(); badPtr[k] = bad[k].data(); nmp[k] = (int32_t)bad[k].size(); }
        Csr b = csr(F.mFeatVec);
        RumiFrameFeatures fv = view(F);
        std::vector<int32_t> matches((size_t)K * F.N, -1), nm(K, 0);
        if (RUMI_GUARDED("ORBmatcher / rumi_search_by_bow_batch", &rumi_facade::no_growth,
                         rumi_search_by_bow_batch(arena(), K, views.data(), fvv.data(), mpPtr.data(), nmp.data(), badPtr.data(), &fv, &b.v, mfNNratio,
                                                  mbCheckOrientation, matches.data(), nm.data())) != RUMI_OK)
            return result;
        for (int k = 0; k < K; k++) {
            std::vector<MapPointT *> &out = vvpMapPointMatches[live[k]];
            out.assign(F.N, static_cast<MapPointT *>(nullptr));
            for (int f = 0; f < F.N; f++) { const int32_t id = matches[(size_t)k * F.N + f]; if (id >= 0) out[f] = mpsKF[k][id]; }
            result[live[k]] = nm[k];
        }
        return result;
    }

    // Search matches between MapPoints seen in KF1 and KF2 by vocabulary node (loop / merge detection)   ORBmatcher.cc:682-804
    template <class KeyFrameT, class MapPointT>
    int SearchByBoW(KeyFrameT *pKF1, KeyFrameT *pKF2, std::vector<MapPointT *> &vpMatches12) {
        const std::vector<MapPointT *> v1 = pKF1->GetMapPointMatches(), v2 = pKF2->GetMapPointMatches();
        // ids: 0..n1-1 = KF1's points, n1.. = KF2's points (only isBad() is needed per id)
        std::vector<int32_t> m1(v1.size(), -1), m2(v2.size(), -1);
        std::vector<uint8_t> bad(v1.size() + v2.size(), 0);
        for (size_t i = 0; i < v1.size(); i++) if (v1[i]) { m1[i] = (int32_t)i; bad[i] = v1[i]->isBad(); }
        for (size_t i = 0; i < v2.size(); i++) if (v2[i]) { m2[i] = (int32_t)(v1.size() + i); bad[v1.size() + i] = v2[i]->isBad(); }
        Csr a = csr(pKF1->mFeatVec), b = csr(pKF2->mFeatVec);
        RumiFrameFeatures k1 = view(*pKF1), k2 = view(*pKF2);
        std::vector<int32_t> m12(pKF1->N, -1);
        int32_t nmatches = 0;
        if (RUMI_GUARDED("ORBmatcher / rumi_search_by_bow_kf", &ORBmatcher::grow_arena, rumi_search_by_bow_kf(arena(), &k1, &a.v, m1.data(), &k2, &b.v, m2.data(), (int)bad.size(), bad.data(), mfNNratio,
                                  mbCheckOrientation, m12.data(), &nmatches)) != RUMI_OK)
            return -1;
        vpMatches12.assign(v1.size(), static_cast<MapPointT *>(nullptr));
        for (size_t i = 0; i < v1.size() && i < m12.size(); i++) if (m12[i] >= 0) vpMatches12[i] = v2[m12[i]];
        return nmatches;
    }

    // Matching to triangulate new MapPoints. Check Epipolar Constraint (LocalMapping::CreateNewMapPoints)  ORBmatcher.cc:806-1013
    // Monocular key-frames.  The fundamental matrix and the epipole are formed here with the reference's own expressions
    // (:812-829 and Pinhole.cpp:109-112), so they are evaluated by the very same Eigen/Sophus code as upstream.
    template <class KeyFrameT>
    int SearchForTriangulation(KeyFrameT *pKF1, KeyFrameT *pKF2, std::vector<std::pair<size_t, size_t>> &vMatchedPairs, const bool bOnlyStereo,
                               const bool bCoarse = false) {
        float F[9], e[2];
#ifdef RUMI_HAVE_SOPHUS
        {
            Sophus::SE3f T1w = pKF1->GetPose(), T2w = pKF2->GetPose(), Tw2 = pKF2->GetPoseInverse();
            Eigen::Vector3f Cw = pKF1->GetCameraCenter();
            Eigen::Vector3f C2 = T2w * Cw;
            Eigen::Vector2f ep = pKF2->mpCamera->project(C2);
            Sophus::SE3f T12 = T1w * Tw2;
            Eigen::Matrix3f R12 = T12.rotationMatrix();
            Eigen::Vector3f t12 = T12.translation();
            Eigen::Matrix3f t12x = Sophus::SO3f::hat(t12);
            Eigen::Matrix3f K1 = pKF1->mpCamera->toK_();
            Eigen::Matrix3f K2 = pKF2->mpCamera->toK_();
            Eigen::Matrix3f F12 = K1.transpose().inverse() * t12x * R12 * K2.inverse();
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) F[r * 3 + c] = F12(r, c);
            e[0] = ep(0); e[1] = ep(1);
        }
#else
        pKF1->EpipolarGeometryTo(pKF2, F, e);            // adapter of the mock data model (tests/cpp/test_facade.cc)
#endif
        const auto v1 = pKF1->GetMapPointMatches(), v2 = pKF2->GetMapPointMatches();
        std::vector<int32_t> m1(pKF1->N, -1), m2(pKF2->N, -1);
        for (size_t i = 0; i < v1.size() && i < m1.size(); i++) if (v1[i]) m1[i] = 0;
        for (size_t i = 0; i < v2.size() && i < m2.size(); i++) if (v2[i]) m2[i] = 0;
        Csr a = csr(pKF1->mFeatVec), b = csr(pKF2->mFeatVec);
        RumiFrameFeatures k1 = view(*pKF1), k2 = view(*pKF2);
        std::vector<int32_t> m12(pKF1->N, -1);
        int32_t nmatches = 0;
        if (RUMI_GUARDED("ORBmatcher / rumi_search_for_triangulation", &ORBmatcher::grow_arena, rumi_search_for_triangulation(arena(), &k1, &a.v, m1.data(), &k2, &b.v, m2.data(), F, e, bOnlyStereo, bCoarse, mbCheckOrientation,
                                          m12.data(), &nmatches)) != RUMI_OK)
            return -1;
        vMatchedPairs.clear();
        vMatchedPairs.reserve(nmatches > 0 ? nmatches : 0);
        for (size_t i = 0; i < m12.size(); i++) if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair(i, (size_t)m12[i]));
        return nmatches;
    }

    // Project MapPoints into KeyFrame and search for duplicated MapPoints (LocalMapping::SearchInNeighbors)  ORBmatcher.cc:1015-1180
    // Monocular key-frames (bRight = false).  The window search runs on the GPU for all points at once; the map mutations are
    // then replayed in list order with the reference's own MapPoint / KeyFrame methods, re-evaluating the isBad / IsInKeyFrame
    // skips at the moment the reference would (a point replaced by an earlier iteration is bad by the time its turn comes).
    template <class KeyFrameT, class MapPointT>
    int Fuse(KeyFrameT *pKF, const std::vector<MapPointT *> &vpMapPoints, const float th = 3.0, const bool bRight = false) {
        if (bRight) {      // second (fisheye) camera of a stereo rig: outside the monocular hot path -- say so instead of fusing against the wrong camera
            rumi_facade::report("ORBmatcher::Fuse", RUMI_E_INVALID, "bRight = true: the second-camera branch (ORBmatcher.cc:1015-1180, mpCamera2) is not built; no point was fused");
            return 0;
        }
        const auto Tcw = pKF->GetPose();
        const auto q = Tcw.unit_quaternion();
        const auto t = Tcw.translation();
        const float T7[7] = {q.x(), q.y(), q.z(), q.w(), t(0), t(1), t(2)};
        float Ow[3];
#ifdef RUMI_HAVE_SOPHUS
        { const Eigen::Vector3f c = pKF->GetCameraCenter(); Ow[0] = c(0); Ow[1] = c(1); Ow[2] = c(2); }
#else
        camera_centre(T7, Ow);
#endif
        std::vector<int32_t> best;
        if (!fuse_search(pKF, T7, Ow, vpMapPoints, [&](MapPointT *p) { return p->isBad() || p->IsInKeyFrame(pKF); }, th, 1, best)) return -1;
        int nFused = 0;
        for (size_t i = 0; i < vpMapPoints.size(); i++) {
            MapPointT *pMP = vpMapPoints[i];
            if (!pMP) continue;
            if (pMP->isBad()) continue;
            else if (pMP->IsInKeyFrame(pKF)) continue;
            if (best[i] < 0) continue;
            MapPointT *pMPinKF = pKF->GetMapPoint(best[i]);                    // :1162-1174
            if (pMPinKF) {
                if (!pMPinKF->isBad()) {
                    if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                    else pMPinKF->Replace(pMP);
                }
            } else {
                pMP->AddObservation(pKF, best[i]);
                pKF->AddMapPoint(pMP, best[i]);
            }
            nFused++;
        }
        return nFused;
    }

#ifdef RUMI_HAVE_SOPHUS
    // Search matches between MapPoints seen in KF1 and KF2 transforming by a Sim3 [s12*R12|t12]              ORBmatcher.cc:1293-1496
    // The camera-frame points are formed with the reference's own Sophus expressions (:1338-1340, :1414-1416).
    template <class KeyFrameT, class MapPointT>
    int SearchBySim3(KeyFrameT *pKF1, KeyFrameT *pKF2, std::vector<MapPointT *> &vpMatches12, const Sophus::Sim3f &S12, const float th) {
        const Sophus::SE3f T1w = pKF1->GetPose(), T2w = pKF2->GetPose();
        const Sophus::Sim3f S21 = S12.inverse();
        const std::vector<MapPointT *> v1 = pKF1->GetMapPointMatches(), v2 = pKF2->GetMapPointMatches();
        const int N1 = (int)v1.size(), N2 = (int)v2.size();
        std::vector<bool> already1(N1, false), already2(N2, false);
        for (int i = 0; i < N1; i++)
            if (MapPointT *p = vpMatches12[i]) {
                already1[i] = true;
                const int idx2 = std::get<0>(p->GetIndexInKeyFrame(pKF2));
                if (idx2 >= 0 && idx2 < N2) already2[idx2] = true;
            }
        struct Side { std::vector<uint8_t> skip, desc; std::vector<float> pc, mn, mx; };
        auto gather = [](const std::vector<MapPointT *> &v, const std::vector<bool> &already, const Sophus::SE3f &Tw, const Sophus::Sim3f &S) {
            Side s;
            const size_t n = v.size();
            s.skip.assign(n, 1); s.desc.assign(n * 32, 0); s.pc.assign(n * 3, 0.f); s.mn.assign(n, 0.f); s.mx.assign(n, 0.f);
            for (size_t i = 0; i < n; i++) {
                MapPointT *p = v[i];
                if (!p || already[i] || p->isBad()) continue;
                s.skip[i] = 0;
                const Eigen::Vector3f p3Dw = p->GetWorldPos();
                const Eigen::Vector3f pa = Tw * p3Dw;
                const Eigen::Vector3f pb = S * pa;
                for (int c = 0; c < 3; c++) s.pc[3 * i + c] = pb(c);
                s.mn[i] = p->GetMinDistance(); s.mx[i] = p->GetMaxDistance();
                const cv::Mat d = p->GetDescriptor();
                std::memcpy(&s.desc[i * 32], d.ptr(0), 32);
            }
            return s;
        };
        const Side a = gather(v1, already1, T1w, S21), b = gather(v2, already2, T2w, S12);
        const float K4[4] = {pKF1->fx, pKF1->fy, pKF1->cx, pKF1->cy};
        RumiFrameFeatures k1 = view(*pKF1), k2 = view(*pKF2);
        k1.n = N1; k2.n = N2;
        std::vector<int32_t> m12(N1 > 0 ? N1 : 1, -1);
        int32_t nFound = 0;
        if (RUMI_GUARDED("ORBmatcher / rumi_search_by_sim3", &ORBmatcher::grow_arena, rumi_search_by_sim3(arena(), &k1, &k2, K4, pKF2->mfLogScaleFactor, a.skip.data(), a.pc.data(), a.mn.data(), a.mx.data(), a.desc.data(),
                                b.skip.data(), b.pc.data(), b.mn.data(), b.mx.data(), b.desc.data(), th, m12.data(), &nFound)) != RUMI_OK)
            return -1;
        for (int i1 = 0; i1 < N1; i1++) if (m12[i1] >= 0) vpMatches12[i1] = v2[m12[i1]];
        return nFound;
    }

    // Project MapPoints into KeyFrame using a given Sim3 and search for duplicated MapPoints               ORBmatcher.cc:1182-1291
    template <class KeyFrameT, class MapPointT>
    int Fuse(KeyFrameT *pKF, Sophus::Sim3f &Scw, const std::vector<MapPointT *> &vpPoints, float th, std::vector<MapPointT *> &vpReplacePoint) {
        const Sophus::SE3f Tcw = Sophus::SE3f(Scw.rotationMatrix(), Scw.translation() / Scw.scale());
        const Eigen::Vector3f Owv = Tcw.inverse().translation();
        const auto q = Tcw.unit_quaternion();
        const float T7[7] = {q.x(), q.y(), q.z(), q.w(), Tcw.translation()(0), Tcw.translation()(1), Tcw.translation()(2)};
        const float Ow[3] = {Owv(0), Owv(1), Owv(2)};
        const std::set<MapPointT *> spAlreadyFound = pKF->GetMapPoints();
        std::vector<int32_t> best;
        if (!fuse_search(pKF, T7, Ow, vpPoints, [&](MapPointT *p) { return p->isBad() || spAlreadyFound.count(p) > 0; }, th, 0, best)) return -1;
        int nFused = 0;
        for (size_t iMP = 0; iMP < vpPoints.size(); iMP++) {
            MapPointT *pMP = vpPoints[iMP];
            if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
            if (best[iMP] < 0) continue;
            MapPointT *pMPinKF = pKF->GetMapPoint(best[iMP]);                  // :1278-1286
            if (pMPinKF) {
                if (!pMPinKF->isBad()) vpReplacePoint[iMP] = pMPinKF;
            } else {
                pMP->AddObservation(pKF, best[iMP]);
                pKF->AddMapPoint(pMP, best[iMP]);
            }
            nFused++;
        }
        return nFused;
    }
#endif

    // Project MapPoints seen in a key-frame into the frame and search matches (Relocalization)            ORBmatcher.cc:1685-1793
    // Needs two one-line accessors on MapPoint: GetMinDistance() / GetMaxDistance() returning mfMinDistance / mfMaxDistance
    // (the reference only exposes the 0.8x / 1.2x invariance values; PredictScale needs the raw one).  INTEGRATION.md §3.
    template <class FrameT, class KeyFrameT, class SetT>
    int SearchByProjection(FrameT &CurrentFrame, KeyFrameT *pKF, const SetT &sAlreadyFound, const float th, const int ORBdist) {
        using MapPointT = typename std::remove_pointer<typename std::decay<decltype(CurrentFrame.mvpMapPoints[0])>::type>::type;
        const std::vector<MapPointT *> vpMPs = pKF->GetMapPointMatches();
        const int nkf = (int)vpMPs.size();
        std::vector<int32_t> kfMp(nkf, -1), curMp(CurrentFrame.N, -1);
        std::vector<uint8_t> skip(nkf, 1), desc((size_t)nkf * 32, 0);
        std::vector<float> pos((size_t)nkf * 3, 0.f), mn(nkf, 0.f), mx(nkf, 0.f);
        for (int i = 0; i < nkf; i++) {
            MapPointT *p = vpMPs[i];
            if (!p) continue;
            kfMp[i] = i;                                              // id = key-frame feature index
            skip[i] = p->isBad() || sAlreadyFound.count(p);
            const auto P = p->GetWorldPos();
            pos[3 * i] = P(0); pos[3 * i + 1] = P(1); pos[3 * i + 2] = P(2);
            mn[i] = p->GetMinDistance(); mx[i] = p->GetMaxDistance();
            const cv::Mat d = p->GetDescriptor();
            std::memcpy(&desc[(size_t)i * 32], d.ptr(0), 32);
        }
        for (int f = 0; f < CurrentFrame.N; f++) if (CurrentFrame.mvpMapPoints[f]) curMp[f] = nkf;      // any non-NULL blocks the feature
        const auto Tcw = CurrentFrame.GetPose();
        const auto q = Tcw.unit_quaternion();
        const auto t = Tcw.translation();
        const float T7[7] = {q.x(), q.y(), q.z(), q.w(), t(0), t(1), t(2)};
        float Ow[3];
        camera_centre(T7, Ow);
        const float K4[4] = {CurrentFrame.fx, CurrentFrame.fy, CurrentFrame.cx, CurrentFrame.cy};
        RumiFrameFeatures cv_ = view(CurrentFrame);
        int32_t nmatches = 0;
        if (RUMI_GUARDED("ORBmatcher / rumi_search_by_projection_reloc", &ORBmatcher::grow_arena, rumi_search_by_projection_reloc(arena(), &cv_, CurrentFrame.mfLogScaleFactor, T7, Ow, K4,
                                            reinterpret_cast<const RumiKeyPoint *>(pKF->mvKeysUn.data()), nkf, kfMp.data(), nkf, skip.data(),
                                            pos.data(), mn.data(), mx.data(), desc.data(), th, ORBdist, mbCheckOrientation, curMp.data(),
                                            &nmatches)) != RUMI_OK)
            return -1;
        for (int f = 0; f < CurrentFrame.N; f++) {
            if (curMp[f] == nkf) continue;                            // untouched pre-existing association
            CurrentFrame.mvpMapPoints[f] = curMp[f] >= 0 ? vpMPs[curMp[f]] : nullptr;
        }
        return nmatches;
    }

#ifdef RUMI_HAVE_SOPHUS
    // Project MapPoints using a Similarity Transformation and search matches (loop detection / merging)   ORBmatcher.cc:372-471
    template <class KeyFrameT, class MapPointT>
    int SearchByProjection(KeyFrameT *pKF, Sophus::Sim3f &Scw, const std::vector<MapPointT *> &vpPoints, std::vector<MapPointT *> &vpMatched,
                           int th, float ratioHamming = 1.0) {
        std::vector<KeyFrameT *> noKFs, noMatchedKF;
        return searchSim3(pKF, Scw, vpPoints, noKFs, vpMatched, noMatchedKF, th, ratioHamming, false);
    }
    // ... and its overload that also reports the key-frame each point came from                            ORBmatcher.cc:473-579
    template <class KeyFrameT, class MapPointT>
    int SearchByProjection(KeyFrameT *pKF, Sophus::Sim3<float> &Scw, const std::vector<MapPointT *> &vpPoints,
                           const std::vector<KeyFrameT *> &vpPointsKFs, std::vector<MapPointT *> &vpMatched,
                           std::vector<KeyFrameT *> &vpMatchedKF, int th, float ratioHamming = 1.0) {
        return searchSim3(pKF, Scw, vpPoints, vpPointsKFs, vpMatched, vpMatchedKF, th, ratioHamming, true);
    }
#endif

protected:
#ifdef RUMI_HAVE_SOPHUS
    template <class KeyFrameT, class MapPointT>
    int searchSim3(KeyFrameT *pKF, Sophus::Sim3f &Scw, const std::vector<MapPointT *> &vpPoints, const std::vector<KeyFrameT *> &vpPointsKFs,
                   std::vector<MapPointT *> &vpMatched, std::vector<KeyFrameT *> &vpMatchedKF, int th, float ratioHamming, bool withKFs) {
        const Sophus::SE3f Tcw = Sophus::SE3f(Scw.rotationMatrix(), Scw.translation() / Scw.scale());      // :380
        const Eigen::Vector3f Owv = Tcw.inverse().translation();
        const auto q = Tcw.unit_quaternion();
        const float T7[7] = {q.x(), q.y(), q.z(), q.w(), Tcw.translation()(0), Tcw.translation()(1), Tcw.translation()(2)};
        const float Ow[3] = {Owv(0), Owv(1), Owv(2)};
        std::set<MapPointT *> spAlreadyFound(vpMatched.begin(), vpMatched.end());
        spAlreadyFound.erase(static_cast<MapPointT *>(nullptr));
        const int nmp = (int)vpPoints.size();
        std::vector<uint8_t> skip(nmp), desc((size_t)nmp * 32);
        std::vector<float> pos((size_t)nmp * 3), nrm((size_t)nmp * 3), mn(nmp), mx(nmp);
        for (int i = 0; i < nmp; i++) {
            MapPointT *p = vpPoints[i];
            skip[i] = p->isBad() || spAlreadyFound.count(p);
            const Eigen::Vector3f P = p->GetWorldPos(), N = p->GetNormal();
            for (int c = 0; c < 3; c++) { pos[3 * i + c] = P(c); nrm[3 * i + c] = N(c); }
            mn[i] = p->GetMinDistance(); mx[i] = p->GetMaxDistance();
            const cv::Mat d = p->GetDescriptor();
            std::memcpy(&desc[(size_t)i * 32], d.ptr(0), 32);
        }
        std::vector<int32_t> matched(pKF->N, -1);
        for (int f = 0; f < pKF->N; f++) if (vpMatched[f]) matched[f] = -2;
        const float K4[4] = {pKF->fx, pKF->fy, pKF->cx, pKF->cy};
        RumiFrameFeatures kv = view(*pKF);
        int32_t nmatches = 0;
        if (RUMI_GUARDED("ORBmatcher / rumi_search_by_projection_sim3", &ORBmatcher::grow_arena, rumi_search_by_projection_sim3(arena(), &kv, pKF->mfLogScaleFactor, T7, Ow, K4, nmp, skip.data(), pos.data(), nrm.data(), mn.data(),
                                           mx.data(), desc.data(), th, ratioHamming, withKFs ? 1 : 0, matched.data(), &nmatches)) != RUMI_OK)
            return -1;
        for (int f = 0; f < pKF->N; f++)
            if (matched[f] >= 0) { vpMatched[f] = vpPoints[matched[f]]; if (withKFs) vpMatchedKF[f] = vpPointsKFs[matched[f]]; }
        return nmatches;
    }
#endif
    // gather + rumi_fuse_candidates.  A point skipped at gather time (NULL, bad, already in the key-frame) stays skipped in the replay
    // (bad is permanent; a point of the key-frame only leaves it by being replaced, i.e. by turning bad).
    template <class KeyFrameT, class MapPointT, class SkipFn>
    static bool fuse_search(KeyFrameT *pKF, const float *T7, const float *Ow, const std::vector<MapPointT *> &pts, SkipFn skipFn, float th, int reproj,
                            std::vector<int32_t> &best) {
        const int nmp = (int)pts.size();
        best.assign(nmp, -1);
        std::vector<uint8_t> skip(nmp, 1), desc((size_t)nmp * 32);
        std::vector<float> pos((size_t)nmp * 3), nrm((size_t)nmp * 3), mn(nmp), mx(nmp);
        for (int i = 0; i < nmp; i++) {
            MapPointT *p = pts[i];
            if (!p || skipFn(p)) continue;
            skip[i] = 0;
            const auto P = p->GetWorldPos(), N = p->GetNormal();
            for (int c = 0; c < 3; c++) { pos[3 * i + c] = P(c); nrm[3 * i + c] = N(c); }
            mn[i] = p->GetMinDistance(); mx[i] = p->GetMaxDistance();
            const cv::Mat d = p->GetDescriptor();
            std::memcpy(&desc[(size_t)i * 32], d.ptr(0), 32);
        }
        const float K4[4] = {pKF->fx, pKF->fy, pKF->cx, pKF->cy};
        RumiFrameFeatures kv = view(*pKF);
        return RUMI_GUARDED("ORBmatcher / rumi_fuse_candidates", &ORBmatcher::grow_arena,
                            rumi_fuse_candidates(arena(), &kv, pKF->mfLogScaleFactor, T7, Ow, K4, nmp, skip.data(), pos.data(), nrm.data(), mn.data(), mx.data(),
                                                 desc.data(), th, reproj, best.data())) == RUMI_OK;
    }
    // Sophus::SE3f::inverse().translation(): conj(q) applied to -t with the same quaternion product form as so3.hpp:358-367
    static void camera_centre(const float *T7, float *Ow) {
        const float qx = -T7[0], qy = -T7[1], qz = -T7[2], qw = T7[3];
        const float p[3] = {T7[4] * -1, T7[5] * -1, T7[6] * -1};
        float u0 = qy * p[2] - qz * p[1], u1 = qz * p[0] - qx * p[2], u2 = qx * p[1] - qy * p[0];
        u0 += u0; u1 += u1; u2 += u2;
        const float c0 = qy * u2 - qz * u1, c1 = qz * u0 - qx * u2, c2 = qx * u1 - qy * u0;
        Ow[0] = (p[0] + qw * u0) + c0; Ow[1] = (p[1] + qw * u1) + c1; Ow[2] = (p[2] + qw * u2) + c2;
    }
    struct Csr {
        std::vector<uint32_t> nodes, idx;
        std::vector<int32_t> off;
        RumiFeatureVector v;
    };
    template <class FeatVecT> static Csr csr(const FeatVecT &fv) {      // DBoW2::FeatureVector = std::map<NodeId, std::vector<unsigned>>
        Csr c;
        c.off.push_back(0);
        for (const auto &kv : fv) {
            c.nodes.push_back((uint32_t)kv.first);
            for (unsigned i : kv.second) c.idx.push_back(i);
            c.off.push_back((int32_t)c.idx.size());
        }
        c.v = RumiFeatureVector{(int32_t)c.nodes.size(), c.nodes.data(), c.off.data(), c.idx.data()};
        return c;
    }
public:   // also used by the free functions of FrameFrustum.h
    template <class FrameT> static RumiFrameFeatures view(const FrameT &F) {
        RumiFrameFeatures v;
        v.n = F.N;
        v.keys_un = reinterpret_cast<const RumiKeyPoint *>(F.mvKeysUn.data());
        v.desc = F.mDescriptors.ptr(0);
        v.min_x = F.mnMinX; v.min_y = F.mnMinY; v.max_x = F.mnMaxX; v.max_y = F.mnMaxY;
        v.scale_factors = F.mvScaleFactors.data();
        v.nlevels = (int32_t)F.mvScaleFactors.size();
        return v;
    }

protected:
    float mfNNratio;
    bool mbCheckOrientation;
};

}  // namespace RUMI_FACADE_NAMESPACE
