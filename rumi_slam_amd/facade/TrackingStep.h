// The Tracking thread's data path on a frame that stays resident on the device (include/rumi_track.h), one reference member function per call:
//   ExtractFrame             Frame::ExtractORB                       R/lib_src/Frame.cc:473-479
//   TrackWithMotionModel     Tracking::TrackWithMotionModel          R/lib_src/Tracking.cc:2441-2518 (monocular, no IMU, after UpdateLastFrame)
//   TrackReferenceKeyFrame   Tracking::TrackReferenceKeyFrame        :2324-2375 (Frame::ComputeBoW on the device)
//   TrackLocalMap            Tracking::TrackLocalMap                 :2520-2607 (after UpdateLocalMap; SearchLocalPoints :2996-3055 inside)
// and TrackFrame, the first and the last fused into ONE device call for a caller that supplies the local set itself.
//
// Why the split.  Tracking::Track builds the local map BETWEEN the two halves: UpdateLocalKeyFrames votes with the matches TrackWithMotionModel
// (or TrackReferenceKeyFrame) left in mCurrentFrame.mvpMapPoints (:3092-3105), so UpdateLocalMap() must run on the host after the first half
// and before TrackLocalMap -- called before, on a frame without matches, it yields an empty local map.  And the three outcomes differ:
//     bOK = TrackWithMotionModel();  if (!bOK) bOK = TrackReferenceKeyFrame();      // :1823-1843: the ONLY fall-back
//     if (bOK) bOK = TrackLocalMap();                                              // :1912-1920: failing here is LOST, not a fall-back
// Reference-side use (Tracking::Track, state OK, velocity valid), with a Frame constructor that skips ExtractORB:
//     rumi_facade::TrackStep st;
//     rumi_facade::ExtractFrame(mCurrentFrame, mImGray, *mpORBextractorLeft);
//     float Tpred[7];  /* mVelocity * mLastFrame.GetPose() as [qx qy qz qw tx ty tz] */
//     bool bOK = rumi_facade::TrackWithMotionModel(mCurrentFrame, mLastFrame, Tpred, 15.f, &st);
//     if (!bOK) bOK = rumi_facade::TrackReferenceKeyFrame(mCurrentFrame, mpReferenceKF, mLastFrame, *mpORBVocabulary, &st);
//     if (bOK) { UpdateLocalMap();                       // the reference's own, unchanged: it reads mCurrentFrame.mvpMapPoints
//                rumi_facade::TrackLocalMap(mCurrentFrame, mvpLocalMapPoints, th, mpLocalMapper->mbFarPoints, mpLocalMapper->mThFarPoints, &st);
//                bOK = st.mnMatchesInliers >= 30; }      // (50 shortly after a relocalisation: the caller's rule, :2589-2607)
// Every function replays on the host exactly what its reference counterpart does to Frame and MapPoint objects, under the reference's own
// conditions: nothing at all when the motion model finds fewer than 20 matches (it returns before optimising), nothing when the BoW search finds
// fewer than 15, the discard loop otherwise; IncreaseVisible / mbTrackInView / IncreaseFound only in TrackLocalMap.
// The discard loop follows the reference's MONOCULAR behaviour to the letter (Tracking.cc:2489-2508: `if (i < mCurrentFrame.Nleft) mbTrackInView =
// false; else mbTrackInViewR = false;` with Nleft = -1, Frame.cc:420): mbTrackInViewR is cleared, mbTrackInView stays as the previous frame's
// SearchLocalPoints left it -- and the next SearchByProjection searches such a point at its OLD mTrackProjX / Y, level and viewing cosine
// (ORBmatcher.cc:46-60).  So the tracking fields ARE part of the replayed state: TrackLocalMap / TrackFrame store mTrackProjX/Y, mnTrackScaleLevel,
// mTrackViewCos, mTrackDepth of the points isInFrustum accepted back into the MapPoints (rumi_track_last_projections) and hand the flags of the
// table's points to the device (RumiTrackPoints.stale_in_view / stale_proj).
#pragma once
#include <algorithm>
#include <cstring>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "ORBextractor.h"
#include "ORBVocabulary.h"
#include "rumi_status.h"
#include "rumi_track.h"

namespace rumi_facade {

struct TrackStep {
    int monoIndex = -1, thMotion = 0, nmatches = 0, ngoodMotion = 0, nmatchesMap = 0, nToMatch = 0, nmatchesLocal = 0, ngoodLocal = 0, mnMatchesInliers = 0;
    int nmatchesBoW = 0;                    // TrackReferenceKeyFrame: SearchByBoW's return value
    bool okMotion = false;                  // what TrackWithMotionModel / TrackReferenceKeyFrame returned (the last one that ran)
    bool ranLocal = false;                  // TrackLocalMap's data path has run (TrackFrame only runs it when okMotion)
    float TcwMotion[7] = {0, 0, 0, 1, 0, 0, 0}, Tcw[7] = {0, 0, 0, 1, 0, 0, 0};
};

// per thread and extractor configuration: the tracker owns an extractor handle, a matcher arena and the result block
inline RumiTracker *&tracker_slot() { thread_local RumiTracker *t = nullptr; return t; }
inline RumiOrbConfig &tracker_cfg() { thread_local RumiOrbConfig c{}; return c; }
inline int &tracker_points() { thread_local int n = 16384; return n; }
// lens distortion of this thread's camera (SetDistortion below): mK and mDistCoef (k1, k2, p1, p2, k3); k1 == 0 = none
struct TrackerDistortion { bool on = false; float K4[4] = {0, 0, 0, 0}, dist5[5] = {0, 0, 0, 0, 0}; };
inline TrackerDistortion &tracker_distortion() { thread_local TrackerDistortion d; return d; }
namespace track_detail_fwd {
inline void TrackerDistortion_set(const float K4[4], const float *distCoef, int nCoef) {
    TrackerDistortion &d = tracker_distortion();
    for (int i = 0; i < 4; i++) d.K4[i] = K4[i];
    for (int i = 0; i < 5; i++) d.dist5[i] = (distCoef && i < nCoef) ? distCoef[i] : 0.f;
    d.on = d.dist5[0] != 0.0f;                               // the reference's own test (Frame.cc:771)
    if (tracker_slot()) { const int rc = rumi_track_set_distortion(tracker_slot(), d.K4, d.on ? d.dist5 : nullptr); if (rc != RUMI_OK) report("rumi_track_set_distortion", rc); }
}
}  // namespace track_detail_fwd

namespace track_detail {

// A table of map points in the order given (each point once), with the per-point data the device stages read
template <class MapPointT> struct PointTable {
    std::unordered_map<const MapPointT *, int> idOf;
    std::vector<MapPointT *> byId;
    std::vector<float> pos, nrm, mn, mx, staleProj;
    std::vector<uint8_t> desc, bad, local, staleIn;
    std::vector<int32_t> obs;
    int id_of(MapPointT *p) {
        auto it = idOf.find(p);
        if (it != idOf.end()) return it->second;
        const int id = (int)byId.size();
        idOf.emplace(p, id); byId.push_back(p);
        return id;
    }
    void fill(int nLocal) {
        const int np = (int)byId.size();
        pos.assign((size_t)np * 3 + 3, 0.f); nrm.assign((size_t)np * 3 + 3, 0.f); mn.assign(np + 1, 0.f); mx.assign(np + 1, 0.f);
        desc.assign((size_t)np * 32 + 32, 0); bad.assign(np + 1, 0); local.assign(np + 1, 0); obs.assign(np + 1, 0);
        staleIn.assign(np + 1, 0); staleProj.assign((size_t)np * 5 + 5, 0.f);
        for (int j = 0; j < np; j++) {
            MapPointT *p = byId[j];
            // what an earlier frame's SearchLocalPoints left in the point (read by the device for discarded outliers only)
            staleIn[j] = p->mbTrackInView ? 1 : 0;
            float *sp = &staleProj[(size_t)j * 5];
            sp[0] = p->mTrackProjX; sp[1] = p->mTrackProjY; sp[2] = (float)p->mnTrackScaleLevel; sp[3] = p->mTrackViewCos; sp[4] = p->mTrackDepth;
            const auto P = p->GetWorldPos(), N = p->GetNormal();
            for (int c = 0; c < 3; c++) { pos[3 * j + c] = P(c); nrm[3 * j + c] = N(c); }
            mn[j] = p->GetMinDistance(); mx[j] = p->GetMaxDistance(); obs[j] = p->Observations(); bad[j] = p->isBad() ? 1 : 0;
            local[j] = j < nLocal ? 1 : 0;
            const cv::Mat d = p->GetDescriptor();
            std::memcpy(&desc[(size_t)j * 32], d.ptr(0), 32);
        }
    }
    RumiTrackPoints view() const {
        return RumiTrackPoints{(int32_t)byId.size(), pos.data(), nrm.data(), mn.data(), mx.data(), desc.data(), obs.data(), bad.data(), local.data(), staleIn.data(), staleProj.data()};
    }
    // Frame::isInFrustum's writes into the MapPoints (Frame.cc:558-630) for the points the device's SearchLocalPoints accepted (in_view == 1)
    void store_projections(RumiTracker *t, const uint8_t *inView) {
        const int np = (int)byId.size();
        if (np == 0) return;
        std::vector<float> pr((size_t)np * 5);
        const int rc = rumi_track_last_projections(t, np, pr.data());
        if (rc != RUMI_OK) { report("rumi_track_last_projections", rc); return; }
        for (int j = 0; j < np; j++) {
            if (inView[j] != 1) continue;
            MapPointT *p = byId[j];
            const float *sp = &pr[(size_t)j * 5];
            p->mTrackProjX = sp[0]; p->mTrackProjY = sp[1]; p->mnTrackScaleLevel = (int)sp[2]; p->mTrackViewCos = sp[3]; p->mTrackDepth = sp[4];
        }
    }
};

// the tracker of this thread (re-created when the extractor's configuration changes or a table outgrows it)
template <class ExtractorT> inline RumiTracker *tracker_for(ExtractorT &extractor, int cols, int rows, int npoints) {
    RumiOrbConfig cfg = extractor.rumiConfig(cols, rows);
    RumiOrbConfig &have = tracker_cfg();
    if (!tracker_slot() || std::memcmp(&cfg, &have, sizeof(cfg)) != 0 || npoints > tracker_points()) {
        if (tracker_slot()) { rumi_track_destroy(tracker_slot()); tracker_slot() = nullptr; }
        while (npoints > tracker_points()) tracker_points() *= 2;
        const int rc = rumi_track_create(&cfg, tracker_points(), -1, &tracker_slot());
        if (rc != RUMI_OK) { report("rumi_track_create", rc); tracker_slot() = nullptr; return nullptr; }
        have = cfg;
        const TrackerDistortion &d = tracker_distortion();
        if (d.on) { const int rd = rumi_track_set_distortion(tracker_slot(), d.K4, d.dist5); if (rd != RUMI_OK) report("rumi_track_set_distortion", rd); }
    }
    return tracker_slot();
}
// a table larger than the tracker: the tracker is re-created, which drops the resident frame -- so tables are sized generously up front and this
// only reports
inline bool table_fits(int npoints, const char *who) {
    if (npoints <= tracker_points()) return true;
    report(who, RUMI_E_CAPACITY);
    return false;
}

template <class FrameT> inline void set_pose(FrameT &F, const float *T7) {
#ifdef RUMI_HAVE_SOPHUS
    F.SetPose(Sophus::SE3f(Eigen::Quaternionf(T7[3], T7[0], T7[1], T7[2]), Eigen::Vector3f(T7[4], T7[5], T7[6])));
#else
    F.SetPoseFromQuatTrans(T7);                             // mock data model of tests/cpp
#endif
}
template <class FrameT> inline void get_pose(const FrameT &F, float *T7) {
    const auto T = F.GetPose();
    const auto q = T.unit_quaternion();
    const auto t = T.translation();
    T7[0] = q.x(); T7[1] = q.y(); T7[2] = q.z(); T7[3] = q.w(); T7[4] = t(0); T7[5] = t(1); T7[6] = t(2);
}

// the frame's features from a device call
template <class FrameT> inline void set_features(FrameT &Cur, std::vector<cv::KeyPoint> &keys, const std::vector<uint8_t> &dsc, int n) {
    keys.resize(n);
    Cur.N = n;
    Cur.mvKeysUn = keys;                                     // distortion-free camera: mvKeysUn == mvKeys (Frame.cc:771-774)
    if (tracker_distortion().on && tracker_slot() && n > 0) {   // Frame::UndistortKeyPoints ran on the device: the resident frame's mvKeysUn
        static_assert(sizeof(cv::KeyPoint) == sizeof(RumiKeyPoint), "cv::KeyPoint must be the 28-byte POD");
        const int rc = rumi_track_undistorted(tracker_slot(), reinterpret_cast<RumiKeyPoint *>(Cur.mvKeysUn.data()), n, nullptr);
        if (rc != RUMI_OK) report("rumi_track_undistorted", rc);
    }
#ifdef RUMI_TRACK_FRAME_HAS_MVKEYS
    Cur.mvKeys = keys;
#endif
    Cur.mDescriptors.create(n, 32, CV_8U);
    for (int i = 0; i < n; i++) std::memcpy(Cur.mDescriptors.ptr(i), &dsc[(size_t)i * 32], 32);
    Cur.mvpMapPoints.assign(n, nullptr); Cur.mvbOutlier.assign(n, false);
}

// "Discard outliers" of TrackWithMotionModel / TrackReferenceKeyFrame (Tracking.cc:2489-2508, 2349-2369) replayed from the device's result
template <class FrameT, class MapPointT>
inline void replay_discard(FrameT &Cur, const std::vector<MapPointT *> &byId, const int32_t *frameMp, const int32_t *discarded) {
    for (int i = 0; i < Cur.N; i++) {
        Cur.mvpMapPoints[i] = frameMp[i] >= 0 ? byId[frameMp[i]] : nullptr;
        Cur.mvbOutlier[i] = false;
        // monocular: `i < mCurrentFrame.Nleft` is false for every i (Nleft = -1), the loop clears mbTrackInViewR and leaves mbTrackInView alone
        if (discarded[i] >= 0) { MapPointT *p = byId[discarded[i]]; p->mbTrackInViewR = false; p->mnLastFrameSeen = Cur.mnId; }
    }
}

}  // namespace track_detail

// The capture target that saves ExtractFrame / TrackFrame their staging copy: a cv::Mat header on the Tracking thread's tracker's pinned staging
// memory (rumi_track_image_buffer).  The camera driver / decoder writes the grey frame into it (or the caller copies it there where it converts
// to grey anyway: Tracking::GrabImageMonocular's cvtColor can take it as destination) and the same Mat is passed on.  Empty Mat on failure.
template <class ExtractorT> inline cv::Mat CaptureBuffer(ExtractorT &extractor, int cols, int rows, int maxPoints = 0) {
    RumiTracker *t = track_detail::tracker_for(extractor, cols, rows, maxPoints);
    uint8_t *buf = nullptr; int32_t stride = 0;
    if (!t || rumi_track_image_buffer(t, cols, rows, &buf, &stride) != RUMI_OK) return cv::Mat();
    return cv::Mat(rows, cols, CV_8UC1, buf, (size_t)stride);
}

// Lens distortion of the Tracking thread's camera: mK (fx, fy, cx, cy) and mDistCoef (k1, k2, p1, p2[, k3]) as Tracking::ParseCamParamFile reads them.
// Call once before the first frame (and again when the calibration changes).  From then on ExtractFrame / TrackFrame fill mvKeysUn with
// Frame::UndistortKeyPoints' result (Frame.cc:770-797) -- computed on the device, where the grid, the searches and PoseOptimization read it too --
// and ImageBounds gives Frame::ComputeImageBounds' mnMinX .. mnMaxY (:799-826) of the resident frame for the Frame's static members.
inline void SetDistortion(const float K4[4], const float *distCoef, int nCoef) {
    track_detail_fwd::TrackerDistortion_set(K4, distCoef, nCoef);
}
inline bool ImageBounds(float &mnMinX, float &mnMinY, float &mnMaxX, float &mnMaxY) {
    float b[4];
    if (!tracker_slot() || rumi_track_undistorted(tracker_slot(), nullptr, 0, b) != RUMI_OK) return false;
    mnMinX = b[0]; mnMinY = b[1]; mnMaxX = b[2]; mnMaxY = b[3];
    return true;
}

// Frame::ExtractORB(0, im, 0, 1000) for a Frame built without it: the features land in the frame AND stay on the device for the calls below.
// Returns monoIndex (-1 on failure, reported through rumi_status.h; the frame stays empty -- no CPU fallback).
template <class FrameT, class ExtractorT> int ExtractFrame(FrameT &Cur, const cv::Mat &im, ExtractorT &extractor, int maxPoints = 0) {
    if (im.empty()) return -1;
    RumiTracker *t = track_detail::tracker_for(extractor, im.cols, im.rows, maxPoints);
    if (!t) return -1;
    const RumiOrbConfig &cfg = tracker_cfg();
    const int cap = cfg.nfeatures + 4 * cfg.nlevels + 64;
    static_assert(sizeof(cv::KeyPoint) == sizeof(RumiKeyPoint), "cv::KeyPoint must be the 28-byte POD");
    std::vector<cv::KeyPoint> keys(cap);
    std::vector<uint8_t> dsc((size_t)cap * 32);
    int32_t n = 0, mono = -1;
    const int rc = rumi_track_extract(t, im.data, im.cols, im.rows, (int)im.step, reinterpret_cast<RumiKeyPoint *>(keys.data()), dsc.data(), cap, &n, &mono);
    if (rc != RUMI_OK) { report("ExtractFrame: rumi_track_extract", rc); return -1; }
    track_detail::set_features(Cur, keys, dsc, n);
    return mono;
}

// Tracking::TrackWithMotionModel on the resident frame.  Returns the reference's return value (false below 20 matches or below 10 map matches).
template <class FrameT> bool TrackWithMotionModel(FrameT &Cur, FrameT &Last, const float *TcwPred7, float th, TrackStep *out) {
    using MapPointT = typename std::remove_pointer<typename std::decay<decltype(Last.mvpMapPoints[0])>::type>::type;
    TrackStep st = out ? *out : TrackStep();
    st.okMotion = false;
    RumiTracker *t = tracker_slot();
    if (!t) { report("TrackWithMotionModel: ExtractFrame has not run on this thread", RUMI_E_INVALID); if (out) *out = st; return false; }
    track_detail::set_pose(Cur, TcwPred7);                                                  // :2454
    std::fill(Cur.mvpMapPoints.begin(), Cur.mvpMapPoints.end(), static_cast<MapPointT *>(nullptr));   // :2457
    track_detail::PointTable<MapPointT> tab;
    std::vector<int32_t> lastMp(Last.N > 0 ? Last.N : 1, -1);
    std::vector<uint8_t> lastOut(Last.N > 0 ? Last.N : 1, 0);
    for (int i = 0; i < Last.N; i++) { if (Last.mvpMapPoints[i]) lastMp[i] = tab.id_of(Last.mvpMapPoints[i]); lastOut[i] = Last.mvbOutlier[i] ? 1 : 0; }
    tab.fill(0);
    if (!track_detail::table_fits(std::max((int)tab.byId.size(), Last.N), "TrackWithMotionModel: more points than the tracker holds (ExtractFrame's maxPoints)")) { if (out) *out = st; return false; }
    std::vector<int32_t> mp((size_t)tracker_cfg().nfeatures + 4 * tracker_cfg().nlevels + 64, -1), dis(mp.size(), -1);
    const float K4[4] = {Cur.fx, Cur.fy, Cur.cx, Cur.cy};
    const RumiTrackPoints P = tab.view();
    RumiTrackResult r;
    const int rc = rumi_track_motion(t, K4, TcwPred7, reinterpret_cast<const RumiKeyPoint *>(Last.mvKeysUn.data()), Last.N, lastMp.data(), lastOut.data(), &P, th,
                                     mp.data(), dis.data(), &r);
    if (rc != RUMI_OK) { report("TrackWithMotionModel: rumi_track_motion", rc); if (out) *out = st; return false; }
    st.monoIndex = r.mono_index; st.thMotion = r.th_motion; st.nmatches = r.nmatches_motion; st.ngoodMotion = r.ngood_motion; st.nmatchesMap = r.nmatches_map;
    std::memcpy(st.TcwMotion, r.Tcw_motion, 28); std::memcpy(st.Tcw, r.Tcw_motion, 28);
    if (r.nmatches_motion < 20) {
        // the function returned before optimising (:2476-2483): the frame holds what the search wrote, nothing else has happened
        for (int i = 0; i < Cur.N; i++) Cur.mvpMapPoints[i] = mp[i] >= 0 ? tab.byId[mp[i]] : nullptr;
        if (out) *out = st;
        return false;
    }
    track_detail::set_pose(Cur, r.Tcw_motion);
    track_detail::replay_discard(Cur, tab.byId, mp.data(), dis.data());
    st.okMotion = r.nmatches_map >= 10;
    if (out) *out = st;
    return st.okMotion;
}

// Tracking::TrackReferenceKeyFrame on the resident frame: Frame::ComputeBoW (mBowVec / mFeatVec filled), SearchByBoW against pRefKF, pose
// initialised with mLastFrame's, PoseOptimization, discard.  Returns the reference's return value.
template <class FrameT, class KeyFrameT>
bool TrackReferenceKeyFrame(FrameT &Cur, KeyFrameT *pRefKF, FrameT &Last, const ORBVocabulary &voc, TrackStep *out, int levelsup = 4) {
    using MapPointT = typename std::remove_pointer<typename std::decay<decltype(Last.mvpMapPoints[0])>::type>::type;
    TrackStep st = out ? *out : TrackStep();
    st.okMotion = false;
    RumiTracker *t = tracker_slot();
    if (!t || !pRefKF) { report("TrackReferenceKeyFrame: ExtractFrame has not run on this thread", RUMI_E_INVALID); if (out) *out = st; return false; }
    const std::vector<MapPointT *> vpKF = pRefKF->GetMapPointMatches();
    track_detail::PointTable<MapPointT> tab;
    std::vector<int32_t> kfMp(vpKF.size() > 0 ? vpKF.size() : 1, -1);
    for (size_t i = 0; i < vpKF.size(); i++) if (vpKF[i]) kfMp[i] = tab.id_of(vpKF[i]);
    tab.fill(0);
    if (!track_detail::table_fits((int)std::max(tab.byId.size(), vpKF.size()), "TrackReferenceKeyFrame: more points than the tracker holds")) { if (out) *out = st; return false; }
    // the key-frame side as the matcher reads it
    RumiFrameFeatures KF{};
    KF.n = (int)pRefKF->mvKeysUn.size();
    KF.keys_un = reinterpret_cast<const RumiKeyPoint *>(pRefKF->mvKeysUn.data());
    KF.desc = pRefKF->mDescriptors.ptr(0);
    KF.min_x = pRefKF->mnMinX; KF.min_y = pRefKF->mnMinY; KF.max_x = pRefKF->mnMaxX; KF.max_y = pRefKF->mnMaxY;
    KF.scale_factors = pRefKF->mvScaleFactors.data(); KF.nlevels = (int)pRefKF->mvScaleFactors.size();
    std::vector<uint32_t> nodes, idx;
    std::vector<int32_t> off(1, 0);
    for (const auto &kv : pRefKF->mFeatVec) { nodes.push_back((uint32_t)kv.first); for (unsigned i : kv.second) idx.push_back(i); off.push_back((int32_t)idx.size()); }
    const RumiFeatureVector kfFv{(int32_t)nodes.size(), nodes.data(), off.data(), idx.data()};
    float Tinit[7];
    track_detail::get_pose(Last, Tinit);
    const size_t capN = (size_t)tracker_cfg().nfeatures + 4 * tracker_cfg().nlevels + 64;
    std::vector<int32_t> mp(capN, -1), dis(capN, -1);
    std::vector<uint32_t> word(capN), node(capN);
    std::vector<double> wgt(capN);
    const float K4[4] = {Cur.fx, Cur.fy, Cur.cx, Cur.cy};
    const RumiTrackPoints P = tab.view();
    RumiTrackResult r;
    const int rc = rumi_track_reference_keyframe(t, voc.handle(), levelsup, K4, Tinit, &KF, &kfFv, kfMp.data(), &P, 0.7f, 1, word.data(), wgt.data(), node.data(),
                                                 mp.data(), dis.data(), &r);
    if (rc != RUMI_OK) { report("TrackReferenceKeyFrame: rumi_track_reference_keyframe", rc); if (out) *out = st; return false; }
    // mCurrentFrame.ComputeBoW(): the two ordered maps from the per-feature transform, in feature order
    {
        const int n = Cur.N;
        std::vector<uint32_t> bowIds(n > 0 ? n : 1), fvNodes(n > 0 ? n : 1), fvIdx(n > 0 ? n : 1);
        std::vector<double> bowVals(n > 0 ? n : 1);
        std::vector<int32_t> fvOff(n + 1);
        int32_t nWords = 0, nNodes = 0;
        if (rumi_voc_assemble(voc.handle(), n, word.data(), wgt.data(), node.data(), bowIds.data(), bowVals.data(), &nWords, fvNodes.data(), fvOff.data(), fvIdx.data(),
                              &nNodes) == RUMI_OK) {
            Cur.mFeatVec.clear();
            for (int a = 0; a < nNodes; a++) {
                auto it = Cur.mFeatVec.insert(Cur.mFeatVec.end(), typename decltype(Cur.mFeatVec)::value_type(fvNodes[a], typename decltype(Cur.mFeatVec)::mapped_type()));
                it->second.assign(fvIdx.begin() + fvOff[a], fvIdx.begin() + fvOff[a + 1]);
            }
#ifdef RUMI_TRACK_FRAME_HAS_BOWVEC
            Cur.mBowVec.clear();
            for (int k = 0; k < nWords; k++) Cur.mBowVec.insert(Cur.mBowVec.end(), typename decltype(Cur.mBowVec)::value_type(bowIds[k], bowVals[k]));
#endif
        }
    }
    st.monoIndex = r.mono_index; st.nmatchesBoW = r.nmatches_motion; st.ngoodMotion = r.ngood_motion; st.nmatchesMap = r.nmatches_map;
    if (r.nmatches_motion < 15) { if (out) *out = st; return false; }       // :2335-2338: vpMapPointMatches is dropped, the frame untouched
    std::memcpy(st.TcwMotion, r.Tcw_motion, 28); std::memcpy(st.Tcw, r.Tcw_motion, 28);
    track_detail::set_pose(Cur, r.Tcw_motion);
    track_detail::replay_discard(Cur, tab.byId, mp.data(), dis.data());
    st.okMotion = r.nmatches_map >= 10;
    if (out) *out = st;
    return st.okMotion;
}

// Tracking::TrackLocalMap after the caller's UpdateLocalMap(): SearchLocalPoints, PoseOptimization, the statistics loop.  Fills
// st->mnMatchesInliers; the decision (>= 30, >= 50 after a relocalisation, ...) is the caller's (:2589-2607).  Returns mnMatchesInliers (-1: error).
template <class FrameT, class MapPointT>
int TrackLocalMap(FrameT &Cur, const std::vector<MapPointT *> &vpLocalMapPoints, float thLocal, bool bFarPoints, float thFarPoints, TrackStep *out) {
    TrackStep st = out ? *out : TrackStep();
    RumiTracker *t = tracker_slot();
    if (!t) { report("TrackLocalMap: ExtractFrame has not run on this thread", RUMI_E_INVALID); return -1; }
    // the table: mvpLocalMapPoints first and in their order (SearchByProjection visits them in that order, and who gets a contested feature
    // depends on it), then the points only the frame holds
    track_detail::PointTable<MapPointT> tab;
    for (MapPointT *p : vpLocalMapPoints) if (p) tab.id_of(p);
    const int nLocal = (int)tab.byId.size();
    std::vector<int32_t> mpIn(Cur.N > 0 ? Cur.N : 1, -1);
    for (int i = 0; i < Cur.N; i++) if (Cur.mvpMapPoints[i]) mpIn[i] = tab.id_of(Cur.mvpMapPoints[i]);
    tab.fill(nLocal);
    const int np = (int)tab.byId.size();
    if (!track_detail::table_fits(np, "TrackLocalMap: more points than the tracker holds (ExtractFrame's maxPoints)")) return -1;
    std::vector<uint8_t> seen(np + 1, 0), inView(np + 1, 0);
    for (int j = 0; j < np; j++) seen[j] = tab.byId[j]->mnLastFrameSeen == Cur.mnId ? 1 : 0;    // the outliers the previous function discarded
    float T7[7];
    track_detail::get_pose(Cur, T7);
    const size_t capN = (size_t)tracker_cfg().nfeatures + 4 * tracker_cfg().nlevels + 64;
    std::vector<int32_t> mp(capN, -1);
    std::vector<uint8_t> outl(capN, 0);
    const float K4[4] = {Cur.fx, Cur.fy, Cur.cx, Cur.cy};
    const RumiTrackPoints P = tab.view();
    RumiTrackResult r;
    const int rc = rumi_track_local(t, K4, T7, mpIn.data(), &P, seen.data(), thLocal, bFarPoints ? 1 : 0, thFarPoints, mp.data(), outl.data(), inView.data(), &r);
    if (rc != RUMI_OK) { report("TrackLocalMap: rumi_track_local", rc); return -1; }
    // ---- the reference's loops over Frame and MapPoint objects, in their order
    for (int i = 0; i < Cur.N; i++) {                        // SearchLocalPoints, first loop (:2998-3010)
        MapPointT *p = Cur.mvpMapPoints[i];
        if (!p) continue;
        if (p->isBad()) { Cur.mvpMapPoints[i] = nullptr; continue; }
        p->IncreaseVisible(); p->mnLastFrameSeen = Cur.mnId; p->mbTrackInView = false;
    }
    for (int j = 0; j < nLocal; j++) {                       // second loop (:3015-3030): isInFrustum sets mbTrackInView either way for the points it sees
        MapPointT *p = tab.byId[j];
        if (p->mnLastFrameSeen == Cur.mnId || p->isBad()) continue;     // (a discarded outlier keeps whatever mbTrackInView it had: inView[j] == 2 if that was true)
        p->mbTrackInView = inView[j] != 0;
        if (inView[j]) p->IncreaseVisible();
    }
    tab.store_projections(t, inView.data());
    st.mnMatchesInliers = 0;
    for (int i = 0; i < Cur.N; i++) {                        // TrackLocalMap (:2573-2586)
        Cur.mvpMapPoints[i] = mp[i] >= 0 ? tab.byId[mp[i]] : nullptr;
        Cur.mvbOutlier[i] = mp[i] >= 0 && outl[i] != 0;
        if (mp[i] >= 0 && !outl[i]) tab.byId[mp[i]]->IncreaseFound();
    }
    track_detail::set_pose(Cur, r.Tcw);
    st.nToMatch = r.n_to_match; st.nmatchesLocal = r.nmatches_local; st.ngoodLocal = r.ngood_local; st.mnMatchesInliers = r.matches_inliers; st.ranLocal = true;
    std::memcpy(st.Tcw, r.Tcw, 28);
    if (out) *out = st;
    return r.matches_inliers;
}

// The fused form: extraction, TrackWithMotionModel and TrackLocalMap in ONE device call, for a caller that supplies the local set ITSELF --
// vpLocalMapPoints must not be the result of an UpdateLocalMap() on this (still empty) frame; the local map of the previous frame's
// TrackLocalMap is the natural choice (the reference rebuilds it from the new matches, which mostly re-elects the same key-frames).
// When the motion model fails by the reference's rules (fewer than 20 matches, or nmatchesMap < 10) TrackLocalMap has NOT run in the reference:
// the fused result is dropped, the first half is re-run on the resident frame through TrackWithMotionModel above (exact state and side effects of
// the failure branch), st->okMotion is false and the caller continues with TrackReferenceKeyFrame -- no re-extraction.  A failing TrackLocalMap
// (st->mnMatchesInliers below the caller's threshold) is LOST in the reference, not a reason to call TrackReferenceKeyFrame.
template <class FrameT, class MapPointT, class ExtractorT>
int TrackFrame(FrameT &Cur, const cv::Mat &im, ExtractorT &extractor, const float *TcwPred7, FrameT &Last, const std::vector<MapPointT *> &vpLocalMapPoints,
               float thMotion, float thLocal, bool bFarPoints, float thFarPoints, TrackStep *out) {
    TrackStep st;
    if (out) *out = st;
    if (im.empty()) return -1;
    // ---- the point table: mvpLocalMapPoints first and in their order, then the points only the last frame holds
    track_detail::PointTable<MapPointT> tab;
    for (MapPointT *p : vpLocalMapPoints) if (p) tab.id_of(p);
    const int nLocal = (int)tab.byId.size();
    std::vector<int32_t> lastMp(Last.N > 0 ? Last.N : 1, -1);
    std::vector<uint8_t> lastOut(Last.N > 0 ? Last.N : 1, 0);
    for (int i = 0; i < Last.N; i++) { if (Last.mvpMapPoints[i]) lastMp[i] = tab.id_of(Last.mvpMapPoints[i]); lastOut[i] = Last.mvbOutlier[i] ? 1 : 0; }
    tab.fill(nLocal);
    const int np = (int)tab.byId.size();
    RumiTracker *t = track_detail::tracker_for(extractor, im.cols, im.rows, std::max(np, Last.N));
    if (!t) return -1;
    const RumiOrbConfig &cfg = tracker_cfg();
    const int cap = cfg.nfeatures + 4 * cfg.nlevels + 64;
    static_assert(sizeof(cv::KeyPoint) == sizeof(RumiKeyPoint), "cv::KeyPoint must be the 28-byte POD");
    std::vector<cv::KeyPoint> keys(cap);
    std::vector<uint8_t> dsc((size_t)cap * 32), outl(cap), inView(np + 1, 0);
    std::vector<int32_t> mpMotion(cap), mpFinal(cap);
    const float K4[4] = {Cur.fx, Cur.fy, Cur.cx, Cur.cy};
    const RumiTrackPoints P = tab.view();
    RumiTrackResult r;
    const int rc = rumi_track_frame(t, im.data, im.cols, im.rows, (int)im.step, K4, TcwPred7, reinterpret_cast<const RumiKeyPoint *>(Last.mvKeysUn.data()), Last.N,
                                    lastMp.data(), lastOut.data(), &P, thMotion, thLocal, bFarPoints ? 1 : 0, thFarPoints, reinterpret_cast<RumiKeyPoint *>(keys.data()),
                                    dsc.data(), cap, mpMotion.data(), mpFinal.data(), outl.data(), inView.data(), &r);
    if (rc != RUMI_OK) { report("TrackFrame: rumi_track_frame", rc); return -1; }    // reported (rumi_status.h); the frame stays empty -- no CPU fallback
    const int n = r.n;
    track_detail::set_features(Cur, keys, dsc, n);
    if (r.nmatches_motion < 20 || r.nmatches_map < 10) {
        // TrackWithMotionModel returned false: the reference has not touched the local map.  Its exact failure state, from the resident frame.
        TrackWithMotionModel(Cur, Last, TcwPred7, thMotion, &st);
        st.monoIndex = r.mono_index;
        if (out) *out = st;
        return r.mono_index;
    }
    // ---- the reference's loops over Frame and MapPoint objects, in their order
    for (int i = 0; i < n; i++) {                            // SearchLocalPoints, first loop (:2998-3010): the matches TrackWithMotionModel kept
        if (mpMotion[i] < 0) continue;
        MapPointT *p = tab.byId[mpMotion[i]];
        p->IncreaseVisible(); p->mnLastFrameSeen = Cur.mnId; p->mbTrackInView = false;
    }
    for (int j = 0; j < nLocal; j++) {                       // second loop (:3015-3030): isInFrustum clears mbTrackInView of every point it tests and sets it
        MapPointT *p = tab.byId[j];                          // again for the ones it accepts.  Not tested: points seen in this frame -- the kept matches (cleared
        if (inView[j] == 2 || p->mnLastFrameSeen == Cur.mnId || p->isBad()) continue;     // above) and the discarded outliers (flag untouched: 2 if it was set) -- and bad points
        p->mbTrackInView = inView[j] == 1;
        if (inView[j] == 1) p->IncreaseVisible();
    }
    tab.store_projections(t, inView.data());
    for (int i = 0; i < n; i++) {                            // TrackLocalMap (:2573-2586)
        if (mpFinal[i] < 0) continue;
        Cur.mvpMapPoints[i] = tab.byId[mpFinal[i]];
        Cur.mvbOutlier[i] = outl[i] != 0;
        if (!outl[i]) tab.byId[mpFinal[i]]->IncreaseFound();
    }
    track_detail::set_pose(Cur, r.Tcw);
    st.monoIndex = r.mono_index; st.thMotion = r.th_motion; st.nmatches = r.nmatches_motion; st.ngoodMotion = r.ngood_motion; st.nmatchesMap = r.nmatches_map;
    st.nToMatch = r.n_to_match; st.nmatchesLocal = r.nmatches_local; st.ngoodLocal = r.ngood_local; st.mnMatchesInliers = r.matches_inliers;
    st.okMotion = true; st.ranLocal = true;
    std::memcpy(st.TcwMotion, r.Tcw_motion, 28); std::memcpy(st.Tcw, r.Tcw, 28);
    if (out) *out = st;
    return r.mono_index;
}

}  // namespace rumi_facade
