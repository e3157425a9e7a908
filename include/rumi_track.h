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
#include "rumi_orb.h"
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

#ifdef __cplusplus
}
#endif
#endif
