// Batched Frame::isInFrustum for Tracking::SearchLocalPoints (R/lib_src/Tracking.cc:3016-3030, R/lib_src/Frame.cc:558-617, mono
// branch): the reference calls `mCurrentFrame.isInFrustum(pMP, 0.5)` once per local map point; this evaluates the whole list
// in one GPU call (include/rumi_match.h: rumi_frame_is_in_frustum) and writes the same tracking fields on every point
// (mbTrackInView, mTrackProjX/Y, mnTrackScaleLevel, mTrackViewCos, mTrackDepth).  result[i] = what isInFrustum returned.
// Needs the raw scale-invariance distances: MapPoint::GetMinDistance() / GetMaxDistance().
//
// Reference-side use, replacing the per-point loop body:
//     std::vector<MapPoint *> cand;                      // points that pass the mnLastFrameSeen / isBad() tests of :3019-3022
//     std::vector<uint8_t> in = rumi_facade::IsInFrustum(mCurrentFrame, cand, 0.5f);
//     for (size_t i = 0; i < cand.size(); i++) { if (in[i]) { cand[i]->IncreaseVisible(); nToMatch++; } ... }
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "ORBmatcher.h"
#include "rumi_status.h"

namespace rumi_facade {

template <class FrameT, class MapPointT>
std::vector<uint8_t> IsInFrustum(FrameT &F, const std::vector<MapPointT *> &vpMPs, float viewingCosLimit) {
    const int n = (int)vpMPs.size();
    std::vector<uint8_t> inView(n > 0 ? n : 1, 0);
    if (n == 0) { inView.clear(); return inView; }
    std::vector<float> pos((size_t)n * 3), nrm((size_t)n * 3), mn(n), mx(n), px(n), py(n), vc(n), depth(n);
    std::vector<int32_t> lvl(n);
    for (int i = 0; i < n; i++) {
        const auto P = vpMPs[i]->GetWorldPos(), N = vpMPs[i]->GetNormal();
        for (int c = 0; c < 3; c++) { pos[3 * i + c] = P(c); nrm[3 * i + c] = N(c); }
        mn[i] = vpMPs[i]->GetMinDistance(); mx[i] = vpMPs[i]->GetMaxDistance();
    }
    float R[9], t[3], Ow[3];
#ifdef RUMI_HAVE_SOPHUS
    {   // Frame::mRcw / mtcw / mOw as UpdatePoseMatrices leaves them (Frame.cc:530-538)
        const Eigen::Matrix3f Rm = F.GetPose().rotationMatrix();
        const Eigen::Vector3f tv = F.GetPose().translation(), ov = F.GetCameraCenter();
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R[r * 3 + c] = Rm(r, c); t[r] = tv(r); Ow[r] = ov(r); }
    }
#else
    F.PoseMatrices(R, t, Ow);                              // adapter of the mock data model (tests/cpp)
#endif
    const float K4[4] = {F.fx, F.fy, F.cx, F.cy};
    if (RUMI_GUARDED("FrameFrustum::IsInFrustum / rumi_frame_is_in_frustum", &RUMI_FACADE_NAMESPACE::ORBmatcher::grow_arena,
                     rumi_frame_is_in_frustum(RUMI_FACADE_NAMESPACE::ORBmatcher::arena(), R, t, Ow, K4, F.mnMinX, F.mnMinY, F.mnMaxX, F.mnMaxY, F.mfLogScaleFactor,
                                              F.mnScaleLevels, viewingCosLimit, n, pos.data(), nrm.data(), mn.data(), mx.data(), inView.data(), px.data(),
                                              py.data(), lvl.data(), vc.data(), depth.data())) != RUMI_OK)
        std::fill(inView.begin(), inView.end(), (uint8_t)0);    // reported (rumi_status.h); no point is in view, as if the frustum were empty -- no CPU fallback
    for (int i = 0; i < n; i++) {
        MapPointT *p = vpMPs[i];
        p->mbTrackInView = inView[i] != 0;
        p->mTrackProjX = px[i]; p->mTrackProjY = py[i];
        if (inView[i]) { p->mnTrackScaleLevel = lvl[i]; p->mTrackViewCos = vc[i]; p->mTrackDepth = depth[i]; }
    }
    return inView;
}

// Tracking::SearchLocalPoints from "int nToMatch = 0" on (R/lib_src/Tracking.cc:3012-3054): the frustum test of every local map point and
// ORBmatcher(0.8).SearchByProjection(mCurrentFrame, mvpLocalMapPoints, th, bFarPoints, thFarPoints) in one GPU call
// (rumi_search_local_points): the per-point tracking fields feed the search on the device and come back once, for the write-back here.
// Returns the match count (0 when nothing is in view: the reference does not search then).  Reference-side use:
//     // ... first loop of SearchLocalPoints over mCurrentFrame.mvpMapPoints unchanged (:2998-3010) ...
//     rumi_facade::SearchLocalPoints(mCurrentFrame, mvpLocalMapPoints, th, mpLocalMapper->mbFarPoints, mpLocalMapper->mThFarPoints);
template <class FrameT, class MapPointT>
int SearchLocalPoints(FrameT &F, const std::vector<MapPointT *> &vpLocalMapPoints, float th, bool bFarPoints, float thFarPoints, float nnratio = 0.8f,
                      int *pnToMatch = nullptr) {
    const int n = (int)vpLocalMapPoints.size();
    if (pnToMatch) *pnToMatch = 0;
    if (n == 0) return 0;
    std::vector<uint8_t> skip(n), desc((size_t)n * 32), inView(n);
    std::vector<float> pos((size_t)n * 3), nrm((size_t)n * 3), mn(n), mx(n), px(n), py(n), vc(n), depth(n);
    std::vector<int32_t> lvl(n), obs(n);
    std::unordered_map<const MapPointT *, int> idOf;
    std::vector<MapPointT *> byId(vpLocalMapPoints);
    for (int i = 0; i < n; i++) {
        MapPointT *p = vpLocalMapPoints[i];
        idOf.emplace(p, i);
        skip[i] = p->mnLastFrameSeen == F.mnId || p->isBad();                              // :3018-3021
        const auto P = p->GetWorldPos(), N = p->GetNormal();
        for (int c = 0; c < 3; c++) { pos[3 * i + c] = P(c); nrm[3 * i + c] = N(c); }
        mn[i] = p->GetMinDistance(); mx[i] = p->GetMaxDistance(); obs[i] = p->Observations();
        const cv::Mat d = p->GetDescriptor();
        std::memcpy(&desc[(size_t)i * 32], d.ptr(0), 32);
    }
    // features that already hold a map point: ids beyond n for points that are not in the local list (skipped, only Observations() is read)
    std::vector<int32_t> frameMp(F.N, -1);
    for (int f = 0; f < F.N; f++) {
        MapPointT *p = F.mvpMapPoints[f];
        if (!p) continue;
        auto it = idOf.find(p);
        if (it != idOf.end()) frameMp[f] = it->second;
        else { frameMp[f] = (int)byId.size(); idOf.emplace(p, (int)byId.size()); byId.push_back(p); obs.push_back(p->Observations()); }
    }
    const int nid = (int)byId.size();
    skip.resize(nid, 1); pos.resize((size_t)nid * 3, 0.f); nrm.resize((size_t)nid * 3, 0.f); mn.resize(nid, 0.f); mx.resize(nid, 0.f);
    desc.resize((size_t)nid * 32, 0); inView.resize(nid); px.resize(nid); py.resize(nid); vc.resize(nid); depth.resize(nid); lvl.resize(nid);
    float R[9], t[3], Ow[3];
#ifdef RUMI_HAVE_SOPHUS
    {
        const Eigen::Matrix3f Rm = F.GetPose().rotationMatrix();
        const Eigen::Vector3f tv = F.GetPose().translation(), ov = F.GetCameraCenter();
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R[r * 3 + c] = Rm(r, c); t[r] = tv(r); Ow[r] = ov(r); }
    }
#else
    F.PoseMatrices(R, t, Ow);
#endif
    const float K4[4] = {F.fx, F.fy, F.cx, F.cy};
    RumiFrameFeatures fv = RUMI_FACADE_NAMESPACE::ORBmatcher::view(F);
    int32_t nToMatch = 0, nmatches = 0;
    if (RUMI_GUARDED("FrameFrustum::SearchLocalPoints / rumi_search_local_points", &RUMI_FACADE_NAMESPACE::ORBmatcher::grow_arena,
                     rumi_search_local_points(RUMI_FACADE_NAMESPACE::ORBmatcher::arena(), &fv, R, t, Ow, K4, F.mfLogScaleFactor, F.mnScaleLevels, 0.5f, nid, skip.data(),
                                              pos.data(), nrm.data(), mn.data(), mx.data(), desc.data(), obs.data(), th, bFarPoints, thFarPoints, nnratio,
                                              inView.data(), px.data(), py.data(), lvl.data(), vc.data(), depth.data(), &nToMatch, frameMp.data(),
                                              &nmatches)) != RUMI_OK)
        return -1;                                          // reported (rumi_status.h); the frame keeps the matches it had -- no CPU fallback
    for (int i = 0; i < n; i++) {
        if (skip[i]) continue;
        MapPointT *p = vpLocalMapPoints[i];
        p->mbTrackInView = inView[i] != 0;                                                  // Frame::isInFrustum's writes (Frame.cc:559-617)
        p->mTrackProjX = px[i]; p->mTrackProjY = py[i];
        if (inView[i]) { p->mnTrackScaleLevel = lvl[i]; p->mTrackViewCos = vc[i]; p->mTrackDepth = depth[i]; p->IncreaseVisible(); }   // :3023-3026
    }
    if (pnToMatch) *pnToMatch = nToMatch;
    if (nToMatch == 0) return 0;
    for (int f = 0; f < F.N; f++) F.mvpMapPoints[f] = frameMp[f] >= 0 ? byId[frameMp[f]] : nullptr;
    return nmatches;
}

}  // namespace rumi_facade
