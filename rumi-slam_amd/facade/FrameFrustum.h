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
#include <vector>

#include "ORBmatcher.h"

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
    if (rumi_frame_is_in_frustum(ORB_SLAM3::ORBmatcher::arena(), R, t, Ow, K4, F.mnMinX, F.mnMinY, F.mnMaxX, F.mnMaxY, F.mfLogScaleFactor,
                                 F.mnScaleLevels, viewingCosLimit, n, pos.data(), nrm.data(), mn.data(), mx.data(), inView.data(), px.data(),
                                 py.data(), lvl.data(), vc.data(), depth.data()) != RUMI_OK) {
        std::fprintf(stderr, "IsInFrustum: %s\n", rumi_last_error());
        std::abort();                                       // no CPU fallback
    }
    for (int i = 0; i < n; i++) {
        MapPointT *p = vpMPs[i];
        p->mbTrackInView = inView[i] != 0;
        p->mTrackProjX = px[i]; p->mTrackProjY = py[i];
        if (inView[i]) { p->mnTrackScaleLevel = lvl[i]; p->mTrackViewCos = vc[i]; p->mTrackDepth = depth[i]; }
    }
    return inView;
}

}  // namespace rumi_facade
