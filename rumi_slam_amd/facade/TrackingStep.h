// One Tracking-thread frame through ONE call (include/rumi_track.h, rumi_track_frame): the data path of
//   Tracking::TrackWithMotionModel  R/lib_src/Tracking.cc:2441-2530 (monocular, no IMU)
//   Tracking::SearchLocalPoints     :2996-3055
//   Tracking::TrackLocalMap         :2545-2607 (after UpdateLocalMap)
// for a Frame whose image has not been extracted yet.  The frame's key-points, descriptors, grid and map-point vector stay on the device between the
// five stages; this function gathers the host-side state once (last frame, the map points it and the local map hold), makes the call and replays
// on the host what the reference's loops do to Frame and MapPoint objects (mvpMapPoints / mvbOutlier, IncreaseVisible, mnLastFrameSeen,
// mbTrackInView, IncreaseFound).  Not replayed, because their only reader is the search that already ran on the device: mTrackProjX/Y,
// mnTrackScaleLevel, mTrackViewCos, mTrackDepth of the points in view, and mnLastFrameSeen of the points whose match was discarded as an outlier.
// The decisions stay with the caller (counts in TrackStep): TrackWithMotionModel returns nmatchesMap >= 10 (and false below 20 matches),
// TrackLocalMap mnMatchesInliers >= 30 (50 shortly after a relocalisation).
//
// Reference-side use (Tracking::Track, state OK, velocity valid), with a Frame constructor that skips ExtractORB (the features arrive here):
//     rumi_facade::TrackStep st;
//     float Tpred[7];  /* mVelocity * mLastFrame.GetPose() as [qx qy qz qw tx ty tz] */
//     UpdateLocalMap();                                     // the local map of the LAST frame's reference key-frame, as TrackLocalMap would build it
//     rumi_facade::TrackFrame(mCurrentFrame, mImGray, *mpORBextractorLeft, Tpred, mLastFrame, mvpLocalMapPoints, 15.f, 1.f,
//                             mpLocalMapper->mbFarPoints, mpLocalMapper->mThFarPoints, &st);
//     bOK = st.nmatches >= 20 && st.nmatchesMap >= 10 && st.mnMatchesInliers >= 30;
#pragma once
#include <cstring>
#include <unordered_map>
#include <vector>

#include "ORBextractor.h"
#include "rumi_status.h"
#include "rumi_track.h"

namespace rumi_facade {

struct TrackStep {
    int monoIndex = -1, thMotion = 0, nmatches = 0, ngoodMotion = 0, nmatchesMap = 0, nToMatch = 0, nmatchesLocal = 0, ngoodLocal = 0, mnMatchesInliers = 0;
    float TcwMotion[7] = {0, 0, 0, 1, 0, 0, 0}, Tcw[7] = {0, 0, 0, 1, 0, 0, 0};
};

// per thread and extractor configuration: the tracker owns an extractor handle, a matcher arena and the result block
inline RumiTracker *&tracker_slot() { thread_local RumiTracker *t = nullptr; return t; }
inline RumiOrbConfig &tracker_cfg() { thread_local RumiOrbConfig c{}; return c; }
inline int &tracker_points() { thread_local int n = 16384; return n; }

template <class FrameT, class MapPointT, class ExtractorT>
int TrackFrame(FrameT &Cur, const cv::Mat &im, ExtractorT &extractor, const float *TcwPred7, FrameT &Last, const std::vector<MapPointT *> &vpLocalMapPoints,
               float thMotion, float thLocal, bool bFarPoints, float thFarPoints, TrackStep *out) {
    TrackStep st;
    if (out) *out = st;
    if (im.empty()) return -1;
    // ---- the point table: mvpLocalMapPoints first and in their order (SearchByProjection visits them in that order, and who gets a contested
    // feature depends on it), then the points only the last frame holds
    std::unordered_map<const MapPointT *, int> idOf;
    std::vector<MapPointT *> byId;
    auto id_of = [&](MapPointT *p) { auto it = idOf.find(p); if (it != idOf.end()) return it->second; const int id = (int)byId.size(); idOf.emplace(p, id); byId.push_back(p); return id; };
    for (MapPointT *p : vpLocalMapPoints) if (p) id_of(p);
    std::vector<uint8_t> local(byId.size(), 1);
    std::vector<int32_t> lastMp(Last.N > 0 ? Last.N : 1, -1);
    std::vector<uint8_t> lastOut(Last.N > 0 ? Last.N : 1, 0);
    for (int i = 0; i < Last.N; i++) { if (Last.mvpMapPoints[i]) lastMp[i] = id_of(Last.mvpMapPoints[i]); lastOut[i] = Last.mvbOutlier[i] ? 1 : 0; }
    const int np = (int)byId.size();
    local.resize(np > 0 ? np : 1, 0);
    std::vector<float> pos((size_t)np * 3 + 3), nrm((size_t)np * 3 + 3), mn(np + 1), mx(np + 1);
    std::vector<uint8_t> desc((size_t)np * 32 + 32), bad(np + 1), inView(np + 1, 0);
    std::vector<int32_t> obs(np + 1);
    for (int j = 0; j < np; j++) {
        MapPointT *p = byId[j];
        const auto P = p->GetWorldPos(), N = p->GetNormal();
        for (int c = 0; c < 3; c++) { pos[3 * j + c] = P(c); nrm[3 * j + c] = N(c); }
        mn[j] = p->GetMinDistance(); mx[j] = p->GetMaxDistance(); obs[j] = p->Observations(); bad[j] = p->isBad() ? 1 : 0;
        const cv::Mat d = p->GetDescriptor();
        std::memcpy(&desc[(size_t)j * 32], d.ptr(0), 32);
    }
    // ---- the tracker of this thread (re-created when the extractor's configuration or the table size outgrows it)
    RumiOrbConfig cfg = extractor.rumiConfig(im.cols, im.rows);
    RumiOrbConfig &have = tracker_cfg();
    if (!tracker_slot() || std::memcmp(&cfg, &have, sizeof(cfg)) != 0 || np > tracker_points() || Last.N > tracker_points()) {
        if (tracker_slot()) { rumi_track_destroy(tracker_slot()); tracker_slot() = nullptr; }
        while (np > tracker_points() || Last.N > tracker_points()) tracker_points() *= 2;
        const int rc = rumi_track_create(&cfg, tracker_points(), -1, &tracker_slot());
        if (rc != RUMI_OK) { report("TrackFrame: rumi_track_create", rc); tracker_slot() = nullptr; return -1; }
        have = cfg;
    }
    const int cap = cfg.nfeatures + 4 * cfg.nlevels + 64;
    static_assert(sizeof(cv::KeyPoint) == sizeof(RumiKeyPoint), "cv::KeyPoint must be the 28-byte POD");
    std::vector<cv::KeyPoint> keys(cap);
    std::vector<uint8_t> dsc((size_t)cap * 32), outl(cap);
    std::vector<int32_t> mpMotion(cap), mpFinal(cap);
    const float K4[4] = {Cur.fx, Cur.fy, Cur.cx, Cur.cy};
    const RumiTrackPoints P{np, pos.data(), nrm.data(), mn.data(), mx.data(), desc.data(), obs.data(), bad.data(), local.data()};
    RumiTrackResult r;
    const int rc = rumi_track_frame(tracker_slot(), im.data, im.cols, im.rows, (int)im.step, K4, TcwPred7, reinterpret_cast<const RumiKeyPoint *>(Last.mvKeysUn.data()), Last.N,
                                    lastMp.data(), lastOut.data(), &P, thMotion, thLocal, bFarPoints ? 1 : 0, thFarPoints, reinterpret_cast<RumiKeyPoint *>(keys.data()),
                                    dsc.data(), cap, mpMotion.data(), mpFinal.data(), outl.data(), inView.data(), &r);
    if (rc != RUMI_OK) { report("TrackFrame: rumi_track_frame", rc); return -1; }    // reported (rumi_status.h); the frame stays empty -- no CPU fallback
    // ---- the frame
    const int n = r.n;
    keys.resize(n);
    Cur.N = n;
    Cur.mvKeysUn = keys;                                     // distortion-free camera: mvKeysUn == mvKeys
#ifdef RUMI_TRACK_FRAME_HAS_MVKEYS
    Cur.mvKeys = keys;
#endif
    Cur.mDescriptors.create(n, 32, CV_8U);
    for (int i = 0; i < n; i++) std::memcpy(Cur.mDescriptors.ptr(i), &dsc[(size_t)i * 32], 32);
    Cur.mvpMapPoints.assign(n, nullptr); Cur.mvbOutlier.assign(n, false);
    // ---- the reference's loops over Frame and MapPoint objects, in their order
    for (int i = 0; i < n; i++) {                            // SearchLocalPoints, first loop (:2998-3010): the matches TrackWithMotionModel kept
        if (mpMotion[i] < 0) continue;
        MapPointT *p = byId[mpMotion[i]];
        p->IncreaseVisible(); p->mnLastFrameSeen = Cur.mnId; p->mbTrackInView = false;
    }
    for (int j = 0; j < np; j++) {                           // second loop (:3015-3030): the local points the frustum test accepted
        if (!inView[j]) continue;
        byId[j]->mbTrackInView = true; byId[j]->IncreaseVisible();
    }
    for (int i = 0; i < n; i++) {                            // TrackLocalMap (:2573-2586)
        if (mpFinal[i] < 0) continue;
        Cur.mvpMapPoints[i] = byId[mpFinal[i]];
        Cur.mvbOutlier[i] = outl[i] != 0;
        if (!outl[i]) byId[mpFinal[i]]->IncreaseFound();
    }
#ifdef RUMI_HAVE_SOPHUS
    Cur.SetPose(Sophus::SE3f(Eigen::Quaternionf(r.Tcw[3], r.Tcw[0], r.Tcw[1], r.Tcw[2]), Eigen::Vector3f(r.Tcw[4], r.Tcw[5], r.Tcw[6])));
#else
    Cur.SetPoseFromQuatTrans(r.Tcw);                         // mock data model of tests/cpp
#endif
    st.monoIndex = r.mono_index; st.thMotion = r.th_motion; st.nmatches = r.nmatches_motion; st.ngoodMotion = r.ngood_motion; st.nmatchesMap = r.nmatches_map;
    st.nToMatch = r.n_to_match; st.nmatchesLocal = r.nmatches_local; st.ngoodLocal = r.ngood_local; st.mnMatchesInliers = r.matches_inliers;
    std::memcpy(st.TcwMotion, r.Tcw_motion, 28); std::memcpy(st.Tcw, r.Tcw, 28);
    if (out) *out = st;
    return r.mono_index;
}

}  // namespace rumi_facade
