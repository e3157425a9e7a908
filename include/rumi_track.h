/* C ABI of one device-resident Tracking step: ORBextractor::operator() -> ORBmatcher::SearchByProjection(Cur, Last) -> Optimizer::PoseOptimization
 * -> Tracking::SearchLocalPoints (Frame::isInFrustum + SearchByProjection(Cur, local points)) -> Optimizer::PoseOptimization, i.e. the data path of
 *   Tracking::TrackWithMotionModel   R/lib_src/Tracking.cc:2441-2530   (monocular, no IMU)
 *   Tracking::TrackLocalMap          R/lib_src/Tracking.cc:2545-2607   (the part after UpdateLocalMap, which stays on the host)
 *   Tracking::SearchLocalPoints      R/lib_src/Tracking.cc:2996-3055
 * in ONE call: the frame's key-points, descriptors, grid and map-point vector (mvpMapPoints) stay in HBM between the five stages; the host
 * stages its inputs once and reads the results once (plus three 16-byte-to-4-KB reads where the reference's control flow needs a number:
 * the feature count for the launch sizes, nmatches for the 2*th retry, and the list-overflow word).
 * The decisions of the two functions (nmatches < 20, nmatchesMap >= 10, mnMatchesInliers < 30, ...) are left to the caller: every number they
 * read is in RumiTrackResult.  Distortion-free pinhole camera (mvKeysUn == mvKeys, image bounds = the image: TUM fr3 as ORB-SLAM3 ships it).
 */
#ifndef RUMI_TRACK_H
#define RUMI_TRACK_H
#include "rumi_match.h"
#include "rumi_orb.h"
#include "rumi_voc.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct RumiTracker RumiTracker;

/* The map points the frame can be matched against, as one table: those the last frame observes (last_mp indexes this table) and
 * mvpLocalMapPoints (local[j] != 0).  The local search visits the table in index order, as SearchByProjection visits mvpLocalMapPoints: list those
 * first and in their order (who gets a contested feature depends on it).  Same per-point data as rumi_search_by_projection_frame /
 * rumi_search_local_points take. */
typedef struct RumiTrackPoints {
    int32_t n;
    const float *pos;        /* [n][3] GetWorldPos() */
    const float *normal;     /* [n][3] GetNormal() */
    const float *min_dist;   /* [n] mfMinDistance (raw; isInFrustum applies 0.8) */
    const float *max_dist;   /* [n] mfMaxDistance (raw; 1.2) */
    const uint8_t *desc;     /* [n][32] GetDescriptor() */
    const int32_t *obs;      /* [n] Observations() */
    const uint8_t *bad;      /* [n] isBad() */
    const uint8_t *local;    /* [n] member of mvpLocalMapPoints */
    /* Optional (NULL = no point carries it): what an EARLIER frame's SearchLocalPoints left in the MapPoint.  The "discard outliers" loop of
     * TrackWithMotionModel / TrackReferenceKeyFrame (Tracking.cc:2489-2508) tests `i < mCurrentFrame.Nleft`; a monocular frame has Nleft = -1
     * (Frame.cc:420), so it clears mbTrackInViewR and leaves mbTrackInView as it was -- SearchLocalPoints then skips the point (mnLastFrameSeen ==
     * mnId) without refreshing it, and SearchByProjection (ORBmatcher.cc:46-60) searches it at its OLD projection.  stale_in_view[i] = the point's
     * mbTrackInView before this frame; stale_proj[i] = {mTrackProjX, mTrackProjY, (float)mnTrackScaleLevel, mTrackViewCos, mTrackDepth}.  Read for
     * discarded outliers only; such a point is searched with these values, does not count in nToMatch and comes back as in_view = 2 (mbTrackInView
     * still set, no IncreaseVisible owed). */
    const uint8_t *stale_in_view;
    const float *stale_proj;
} RumiTrackPoints;

typedef struct RumiTrackResult {
    int32_t n, mono_index;        /* ORBextractor::operator(): number of key-points, monoIndex */
    int32_t th_motion;            /* search radius SearchByProjection(Cur, Last) ended with: th or 2 * th (Tracking.cc:2469-2474) */
    int32_t nmatches_motion;      /* its return value */
    int32_t ngood_motion;         /* Optimizer::PoseOptimization return value (0: not run, fewer than 20 matches) */
    int32_t nmatches_map;         /* nmatchesMap (Tracking.cc:2489-2508) */
    int32_t n_to_match;           /* nToMatch of SearchLocalPoints */
    int32_t nmatches_local;       /* SearchByProjection(Cur, local points) return value */
    int32_t ngood_local;          /* second PoseOptimization return value */
    int32_t matches_inliers;      /* mnMatchesInliers (Tracking.cc:2573-2586, mbOnlyTracking == false) */
    float Tcw_motion[7];          /* pose after TrackWithMotionModel: [qx qy qz qw tx ty tz] */
    float Tcw[7];                 /* pose after TrackLocalMap */
    float Rcw[9], tcw[3], Ow[3];  /* Frame::UpdatePoseMatrices of Tcw_motion, as SearchLocalPoints' isInFrustum used them */
} RumiTrackResult;

/* cfg: the extractor's configuration (max_width / max_height = the camera); max_points: largest RumiTrackPoints.n / last-frame feature count. */
int rumi_track_create(const RumiOrbConfig *cfg, int32_t max_points, int32_t device, RumiTracker **out);
void rumi_track_destroy(RumiTracker *t);

/* The tracker's pinned staging memory for a w x h frame (*stride = w rounded up to 4 bytes per row).  Optional: a caller whose camera driver or
 * decoder writes frames straight into it (the reference: the cv::Mat Tracking::GrabImageMonocular receives, constructed on this memory) and then
 * passes this pointer and stride as `img` / `stride` below saves the staging copy of rumi_track_frame / rumi_track_extract (~20 us for 640 x 480). */
int rumi_track_image_buffer(RumiTracker *t, int32_t w, int32_t h, uint8_t **buf, int32_t *stride);

/* One frame.  img: host, 8-bit grey, `stride` bytes per row.  Tcw_pred7 = mVelocity * mLastFrame.GetPose().  last_*: mLastFrame.mvKeysUn, the
 * table index of mLastFrame.mvpMapPoints[i] (-1: none) and mLastFrame.mvbOutlier.  th_motion = 15 (mono), th_local as SearchLocalPoints
 * chooses it (1 by default), far_points / th_far_points = mpLocalMapper->mbFarPoints / mThFarPoints.
 * Outputs: keys_out / desc_out [cap] (+ res->n): the frame's features; frame_mp_motion [cap]: mvpMapPoints as table indices after
 * TrackWithMotionModel's outlier removal; frame_mp [cap] / outlier [cap]: mvpMapPoints and mvbOutlier after TrackLocalMap; in_view [pts->n]:
 * mbTrackInView of every table point after SearchLocalPoints (the points whose IncreaseVisible the caller owes).
 * Returns RUMI_OK also when tracking "fails" by the reference's rules (read the counts); RUMI_E_* on errors. */
int rumi_track_frame(RumiTracker *t, const uint8_t *img, int32_t w, int32_t h, int32_t stride, const float *K4, const float *Tcw_pred7,
                     const RumiKeyPoint *last_keys_un, int32_t nlast, const int32_t *last_mp, const uint8_t *last_outlier,
                     const RumiTrackPoints *pts, float th_motion, float th_local, int32_t far_points, float th_far_points,
                     RumiKeyPoint *keys_out, uint8_t *desc_out, int32_t cap, int32_t *frame_mp_motion, int32_t *frame_mp, uint8_t *outlier,
                     uint8_t *in_view, RumiTrackResult *res);

/* ---- step-wise entries -------------------------------------------------------------------------------------------------------------------
 * The same stages ONE member function of Tracking at a time, for a host that keeps the reference's control flow: Tracking::Track decides
 * between TrackWithMotionModel and TrackReferenceKeyFrame (Tracking.cc:1823-1843), UpdateLocalMap builds the local set from the matches those
 * leave in mCurrentFrame.mvpMapPoints (:3092-3105) -- so it has to run BETWEEN them and TrackLocalMap, on the host -- and a failed
 * TrackLocalMap is LOST, not a fall-back.  The frame extracted by rumi_track_extract stays resident on the device (key-points, descriptors,
 * grid, and after rumi_track_reference_keyframe its FeatureVector) for every later call; each call takes the point table it needs and
 * returns what its reference function leaves in the Frame.  rumi_track_frame above remains the fused form for a caller that supplies the
 * local set itself. */

/* Frame::ExtractORB(0, im, 0, 1000) (Frame.cc:473-479): host image in, key-points / descriptors out (keys_out, desc_out [cap >= nfeatures +
 * 4 nlevels + 64]) and kept on the device. */
int rumi_track_extract(RumiTracker *t, const uint8_t *img, int32_t w, int32_t h, int32_t stride, RumiKeyPoint *keys_out, uint8_t *desc_out,
                       int32_t cap, int32_t *n_out, int32_t *mono_out);

/* Tracking::TrackWithMotionModel (Tracking.cc:2441-2518), monocular, no IMU, after UpdateLastFrame: SearchByProjection(Cur, Last, th) with
 * the 2 th retry; with >= 20 matches PoseOptimization and the "discard outliers" loop.  pts: the table last_mp indexes (pos, desc, obs, bad
 * are read).  Outputs: frame_mp [cap] = mCurrentFrame.mvpMapPoints as table indices when the function returns; discarded [cap] = the table
 * index of the point whose match was discarded as an outlier at that feature (-1 elsewhere): the caller owes those points
 * mbTrackInView = false and mnLastFrameSeen = mCurrentFrame.mnId.  res: n, mono_index, th_motion, nmatches_motion (< 20: the function
 * returned false before optimising; frame_mp holds the search's result), ngood_motion, nmatches_map, Tcw_motion (= Tcw). */
int rumi_track_motion(RumiTracker *t, const float *K4, const float *Tcw_pred7, const RumiKeyPoint *last_keys_un, int32_t nlast,
                      const int32_t *last_mp, const uint8_t *last_outlier, const RumiTrackPoints *pts, float th_motion, int32_t *frame_mp,
                      int32_t *discarded, RumiTrackResult *res);

/* Tracking::TrackReferenceKeyFrame (Tracking.cc:2324-2375): Frame::ComputeBoW (Frame.cc:763-768: transform(..., levelsup = 4), tree descent and
 * FeatureVector on the device), ORBmatcher(nnratio = 0.7, true).SearchByBoW(mpReferenceKF, mCurrentFrame, vpMapPointMatches); with >= 15
 * matches SetPose(Tcw_init7 = mLastFrame.GetPose()), PoseOptimization and the "discard outliers" loop.
 * KF / kf_fv / kf_mp: the reference key-frame's undistorted key-points + descriptors, its FeatureVector (CSR, rumi_match.h) and
 * GetMapPointMatches() as indices into pts (-1 none); pts: that key-frame's map points (pos, desc unused, obs, bad).
 * Outputs: word_id / word_weight / node_id [cap]: the per-feature transform, from which rumi_voc_assemble gives mBowVec and mFeatVec;
 * frame_mp, discarded as rumi_track_motion; res->nmatches_motion = SearchByBoW's return value (< 15: the function returned false, frame_mp
 * holds vpMapPointMatches, which the reference does not assign to the frame then), ngood_motion, nmatches_map, Tcw_motion. */
int rumi_track_reference_keyframe(RumiTracker *t, RumiVocabulary *voc, int32_t levelsup, const float *K4, const float *Tcw_init7,
                                  const RumiFrameFeatures *KF, const RumiFeatureVector *kf_fv, const int32_t *kf_mp, const RumiTrackPoints *pts,
                                  float nnratio, int32_t check_orientation, uint32_t *word_id, double *word_weight, uint32_t *node_id,
                                  int32_t *frame_mp, int32_t *discarded, RumiTrackResult *res);

/* Tracking::TrackLocalMap after UpdateLocalMap (Tracking.cc:2520-2607) = SearchLocalPoints (:2996-3055) + PoseOptimization + the statistics
 * loop.  Tcw7 = mCurrentFrame.GetPose(); frame_mp_in [n] = mCurrentFrame.mvpMapPoints as indices into pts (the table: mvpLocalMapPoints first
 * and in their order with local = 1, then any other point the frame holds); seen_in [pts->n] (may be NULL): points with mnLastFrameSeen ==
 * mCurrentFrame.mnId already (the outliers the previous function discarded).  Outputs as rumi_track_frame: frame_mp / outlier [cap],
 * in_view [pts->n]; res: n_to_match, nmatches_local, ngood_local, matches_inliers, Tcw, Rcw / tcw / Ow of Tcw7. */
int rumi_track_local(RumiTracker *t, const float *K4, const float *Tcw7, const int32_t *frame_mp_in, const RumiTrackPoints *pts,
                     const uint8_t *seen_in, float th_local, int32_t far_points, float th_far_points, int32_t *frame_mp, uint8_t *outlier,
                     uint8_t *in_view, RumiTrackResult *res);

/* Lens distortion (Frame::UndistortKeyPoints, Frame::ComputeImageBounds: R/lib_src/Frame.cc:770-826, cv::undistortPoints(mat, mat, K, mDistCoef,
 * cv::Mat(), mK): 5 fixed-point iterations of the inverse radial-tangential model in double, then P = K; OpenCV's arithmetic restated, parity
 * unpinned).  K4 = fx, fy, cx, cy of mK, dist5 = (k1, k2, p1, p2, k3) (R/config/euroc_ori.yaml:23-31: k1 = -0.283).  From the next
 * rumi_track_extract / rumi_track_frame on, the resident frame carries mvKeysUn -- the grid (AssignFeaturesToGrid), every search and
 * PoseOptimization read those -- and the bounds mnMinX .. mnMaxY of the undistorted corners; keys_out of those calls stays mvKeys.
 * dist5 == NULL or dist5[0] == 0 switches it off (the reference's own test, Frame.cc:771). */
int rumi_track_set_distortion(RumiTracker *t, const float *K4, const float *dist5);
/* mvKeysUn of the resident frame (keys_un_out [cap >= n], may be NULL) and bounds4 = {mnMinX, mnMinY, mnMaxX, mnMaxY} (may be NULL). */
int rumi_track_undistorted(RumiTracker *t, RumiKeyPoint *keys_un_out, int32_t cap, float *bounds4);

/* mTrackProjX, mTrackProjY, (float)mnTrackScaleLevel, mTrackViewCos, mTrackDepth of every table point, proj5_out [n_points][5], as the
 * SearchLocalPoints of the LAST rumi_track_frame / rumi_track_local call left them (Frame::isInFrustum writes these into the MapPoint, Frame.cc:558-630;
 * meaningful for points with in_view = 1).  A caller that keeps MapPoint objects (the facade) stores them back: a later frame's discarded outlier is
 * searched at these values (RumiTrackPoints.stale_proj).  One more device-to-host copy; valid until the next rumi_track_* call on the tracker. */
int rumi_track_last_projections(RumiTracker *t, int32_t n_points, float *proj5_out);

#ifdef __cplusplus
}
#endif
#endif
