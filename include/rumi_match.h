/*
 * rumi_match.h — C ABI of the MI355X-native Hamming matchers (librumi_hip.so).
 *
 * Drop-in boundary for ORB_SLAM3::ORBmatcher (R/ = /root/reference/src/rumi-slam/):
 *   R/include/cloud_edge_slam_lib/ORBmatcher.h:36-103   class surface (ctor nnratio/checkOri, TH_LOW/TH_HIGH/HISTO_LENGTH)
 *   R/lib_src/ORBmatcher.cc:39-196      SearchByProjection(Frame&, const vector<MapPoint*>&, th, bFarPoints, thFarPoints)
 *   R/lib_src/ORBmatcher.cc:198-370     SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&)
 *   R/lib_src/ORBmatcher.cc:1498-1683   SearchByProjection(Frame& Cur, const Frame& Last, th, bMono)
 *   R/lib_src/ORBmatcher.cc:372-579     SearchByProjection(KeyFrame*, Sim3f&, points[, pointKFs], matched[, matchedKF], th, ratioHamming)
 *   R/lib_src/ORBmatcher.cc:1685-1793   SearchByProjection(Frame&, KeyFrame*, set<MapPoint*>&, th, ORBdist)
 *   R/lib_src/ORBmatcher.cc:682-804     SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&)
 *   R/lib_src/ORBmatcher.cc:1830-1844   DescriptorDistance
 *   R/lib_src/Frame.cc:441-466,695-761  AssignFeaturesToGrid / GetFeaturesInArea / PosInGrid (candidate order)
 * The facade marshals Frame / KeyFrame / MapPoint pointers into the flat views below: a MapPoint* becomes an
 * int32 id into caller-side arrays, NULL becomes -1.  Mono branches only (the node hard-codes MONOCULAR,
 * R/src/cloud_edge_main.cpp:267).  Results are bit-identical to the reference's sequential loops, including
 * the "feature already taken by an earlier map point" side effect and the first-candidate-wins tie rule.
 *
 * Status codes, error string and threading rules: rumi_orb.h.
 */
#ifndef RUMI_MATCH_H
#define RUMI_MATCH_H

#include <stdint.h>

#include "rumi_orb.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RUMI_TH_HIGH 100      /* ORBmatcher::TH_HIGH      ORBmatcher.cc:30 */
#define RUMI_TH_LOW 50        /* ORBmatcher::TH_LOW       ORBmatcher.cc:31 */
#define RUMI_HISTO_LENGTH 30  /* ORBmatcher::HISTO_LENGTH ORBmatcher.cc:32 */

/* What the matchers read out of a Frame (or KeyFrame): host pointers. */
typedef struct RumiFrameFeatures {
    int32_t n;                    /* Frame::N */
    const RumiKeyPoint *keys_un;  /* Frame::mvKeysUn (pt, angle, octave are used) */
    const uint8_t *desc;          /* Frame::mDescriptors, n x 32 */
    float min_x, min_y, max_x, max_y;   /* Frame::mnMinX ... mnMaxY (image bounds) */
    const float *scale_factors;   /* Frame::mvScaleFactors */
    int32_t nlevels;
} RumiFrameFeatures;

/* DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned>>) in CSR form, node ids ascending. */
typedef struct RumiFeatureVector {
    int32_t n_nodes;
    const uint32_t *node_ids;     /* [n_nodes] ascending */
    const int32_t *offsets;       /* [n_nodes + 1] */
    const uint32_t *indices;      /* [offsets[n_nodes]] feature indices, in the vectors' order */
} RumiFeatureVector;

typedef struct RumiMatcher RumiMatcher;

/* static int ORBmatcher::DescriptorDistance(const cv::Mat&, const cv::Mat&) — host, no device needed. */
int rumi_descriptor_distance(const uint8_t *a32, const uint8_t *b32);

/* Scratch arenas for frames of up to max_features features and calls of up to max_queries queries
 * (map points / last-frame features / key-frame features). */
int rumi_match_create(int32_t max_features, int32_t max_queries, int32_t device, RumiMatcher **out);
void rumi_match_destroy(RumiMatcher *m);

/* SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th, bFarPoints, thFarPoints) — TrackLocalMap.
 * Per map point i (arrays of length nmp): the fields Frame::isInFrustum left on it (mbTrackInView, mTrackProjX/Y,
 * mnTrackScaleLevel, mTrackViewCos, mTrackDepth), isBad(), GetDescriptor() (nmp x 32) and Observations().
 * frame_mp [F->n] in/out = F.mvpMapPoints as ids into the same map-point arrays (-1 = NULL).
 * *nmatches_out = the reference's return value. */
int rumi_search_by_projection_mappoints(RumiMatcher *m, const RumiFrameFeatures *F, int32_t nmp,
                                        const uint8_t *track_in_view, const float *proj_x, const float *proj_y,
                                        const int32_t *scale_level, const float *view_cos, const float *track_depth,
                                        const uint8_t *is_bad, const uint8_t *mp_desc, const int32_t *mp_obs, float th,
                                        int32_t far_points, float th_far_points, float nnratio, int32_t *frame_mp,
                                        int32_t *nmatches_out);

/* SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono = true) — TrackWithMotionModel.
 * Tcw7 = CurrentFrame.GetPose() as Sophus stores it: unit quaternion (x,y,z,w) then translation; K4 = fx,fy,cx,cy.
 * last_mp[nlast] = LastFrame.mvpMapPoints as ids (-1 NULL) into mp_pos (nmp x 3, GetWorldPos), mp_desc (nmp x 32),
 * mp_obs (Observations()); last_outlier = LastFrame.mvbOutlier.  cur_mp [Cur->n] in/out = CurrentFrame.mvpMapPoints. */
int rumi_search_by_projection_frame(RumiMatcher *m, const RumiFrameFeatures *Cur, const float *Tcw7, const float *K4,
                                    const RumiKeyPoint *last_keys, int32_t nlast, const int32_t *last_mp,
                                    const uint8_t *last_outlier, int32_t nmp, const float *mp_pos, const uint8_t *mp_desc,
                                    const int32_t *mp_obs, float th, int32_t check_orientation, int32_t *cur_mp,
                                    int32_t *nmatches_out);

/* SearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches) — TrackReferenceKeyFrame / Relocalization.
 * kf_mp[KF->n] = pKF->GetMapPointMatches() as ids (-1 NULL); mp_bad[nmp] = isBad().  matches [F->n] out (-1 = NULL). */
int rumi_search_by_bow(RumiMatcher *m, const RumiFrameFeatures *KF, const RumiFeatureVector *kf_fv, const int32_t *kf_mp,
                       int32_t nmp, const uint8_t *mp_bad, const RumiFrameFeatures *F, const RumiFeatureVector *f_fv,
                       float nnratio, int32_t check_orientation, int32_t *matches, int32_t *nmatches_out);

/* SearchByBoW(KF_k, F, matches_k) for K candidate key-frames against ONE frame in a single launch: what Tracking::Relocalization does candidate by
 * candidate (Tracking.cc:3240-3260; each walk starts from an empty vpMapPointMatches, so the K searches are independent).  KFs / kf_fvs: arrays of K;
 * kf_mp[k][i]: map-point index of key-frame k's feature i in ITS OWN numbering (-1 none), nmp[k] / mp_bad[k]: that numbering's size and bad
 * flags (mp_bad[k] may be NULL).  matches [K][F->n]: the map-point index (key-frame k's numbering) matched to every frame feature, -1 none;
 * nmatches_out [K].  Results are those of K rumi_search_by_bow calls. */
int rumi_search_by_bow_batch(RumiMatcher *m, int32_t K, const RumiFrameFeatures *KFs, const RumiFeatureVector *kf_fvs, const int32_t *const *kf_mp,
                             const int32_t *nmp, const uint8_t *const *mp_bad, const RumiFrameFeatures *F, const RumiFeatureVector *f_fv,
                             float nnratio, int32_t check_orientation, int32_t *matches, int32_t *nmatches_out);

/* SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12) — loop / merge detection (ORBmatcher.cc:682-804).
 * kf1_mp / kf2_mp: GetMapPointMatches() of the two key-frames as ids (-1 NULL) into mp_bad[nmp].
 * matches12[KF1->n] out: index of the KF2 FEATURE whose map point the reference stores in vpMatches12[idx1] (-1 = NULL). */
int rumi_search_by_bow_kf(RumiMatcher *m, const RumiFrameFeatures *KF1, const RumiFeatureVector *fv1, const int32_t *kf1_mp,
                          const RumiFrameFeatures *KF2, const RumiFeatureVector *fv2, const int32_t *kf2_mp, int32_t nmp,
                          const uint8_t *mp_bad, float nnratio, int32_t check_orientation, int32_t *matches12,
                          int32_t *nmatches_out);

/* SearchByProjection(KeyFrame *pKF, Sophus::Sim3f &Scw, const vector<MapPoint*> &vpPoints, vector<MapPoint*> &vpMatched, int th,
 * float ratioHamming)                                        (ORBmatcher.cc:372-471, explicit_invz = 0) and its overload with
 * vpPointsKFs / vpMatchedKF                                  (ORBmatcher.cc:473-579, explicit_invz = 1: u = fx * (x * (1/z)) + cx).
 * Tcw7 / Ow3: the SE3 and camera centre the reference derives from Scw (:380-381), computed by the facade with Sophus.
 * Per candidate point i (nmp): skip = isBad() || already in vpMatched; GetWorldPos, GetNormal, mfMinDistance, mfMaxDistance,
 * GetDescriptor.  log_scale_factor = pKF->mfLogScaleFactor.
 * matched[KF->n] in/out: -1 = free, any other value = the feature already holds a point; on return newly matched features
 * hold the INDEX i of their point (the facade maps it to vpPoints[i] / vpPointsKFs[i]). */
int rumi_search_by_projection_sim3(RumiMatcher *m, const RumiFrameFeatures *KF, float log_scale_factor, const float *Tcw7,
                                   const float *Ow3, const float *K4, int32_t nmp, const uint8_t *skip, const float *mp_pos,
                                   const float *mp_normal, const float *mp_min_dist, const float *mp_max_dist, const uint8_t *mp_desc,
                                   int32_t th, float ratio_hamming, int32_t explicit_invz, int32_t *matched, int32_t *nmatches_out);

/* SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, const float th, const int ORBdist)
 * — Relocalization (ORBmatcher.cc:1685-1793; Tracking.cc:3315,3327).  kf_mp[nkf] = pKF->GetMapPointMatches() as ids into the
 * per-point arrays (skip = isBad() || sAlreadyFound.count()); Ow3 = camera centre of CurrentFrame.  cur_mp in/out. */
int rumi_search_by_projection_reloc(RumiMatcher *m, const RumiFrameFeatures *Cur, float log_scale_factor, const float *Tcw7,
                                    const float *Ow3, const float *K4, const RumiKeyPoint *kf_keys, int32_t nkf, const int32_t *kf_mp,
                                    int32_t nmp, const uint8_t *skip, const float *mp_pos, const float *mp_min_dist,
                                    const float *mp_max_dist, const uint8_t *mp_desc, float th, int32_t orb_dist,
                                    int32_t check_orientation, int32_t *cur_mp, int32_t *nmatches_out);

/* ORBmatcher::SearchForInitialization(Frame &F1, Frame &F2, vector<cv::Point2f> &vbPrevMatched, vector<int> &vnMatches12,
 * int windowSize) — R/lib_src/ORBmatcher.cc:581-680 (monocular initialisation, Tracking.cc:2115).
 * prev_matched [F1->n][2] is read and updated in place (:675-677); matches12 [F1->n] receives vnMatches12. */
int rumi_search_for_initialization(RumiMatcher *m, const RumiFrameFeatures *F1, const RumiFrameFeatures *F2, float *prev_matched,
                                   int32_t window_size, float nnratio, int32_t check_orientation, int32_t *matches12,
                                   int32_t *nmatches_out);

/* ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, vector<pair<size_t,size_t>> &vMatchedPairs, bOnlyStereo,
 * bCoarse) — R/lib_src/ORBmatcher.cc:806-1013, monocular branch (LocalMapping::CreateNewMapPoints, LocalMapping.cc:425).
 * kf*_mp: >= 0 where the key-frame already holds a map point for the feature.  F12 (row-major 3x3) is the fundamental matrix
 * K1^-T [t12]x R12 K2^-1 that Pinhole::epipolarConstrain (Pinhole.cpp:107-129) rebuilds for every pair, epipole2 =
 * pKF2->mpCamera->project(T2w * pKF1->GetCameraCenter()) (:815-818): both are formed by the caller with the reference's own
 * Eigen/Sophus expressions (facade/ORBmatcher.h), so no Eigen arithmetic is restated on this side of the ABI.
 * matches12 [KF1->n] = vMatches12 (vMatchedPairs = its non-negative entries in index order). */
int rumi_search_for_triangulation(RumiMatcher *m, const RumiFrameFeatures *KF1, const RumiFeatureVector *fv1, const int32_t *kf1_mp,
                                  const RumiFrameFeatures *KF2, const RumiFeatureVector *fv2, const int32_t *kf2_mp, const float *F12,
                                  const float *epipole2, int32_t only_stereo, int32_t coarse, int32_t check_orientation,
                                  int32_t *matches12, int32_t *nmatches_out);

/* The search half of both ORBmatcher::Fuse overloads — R/lib_src/ORBmatcher.cc:1015-1180 (KeyFrame*, points, th, bRight=false;
 * LocalMapping::SearchInNeighbors, LocalMapping.cc:699-727) and :1182-1291 (KeyFrame*, Sim3f&, points, th, vpReplacePoint;
 * LoopClosing / CloudMerging): for every map point, the key-frame feature it fuses with (best_idx, -1 = none: skipped, not
 * visible, or best Hamming distance > TH_LOW).  The points do not compete for features, so this part is data-parallel; the
 * map mutations that follow (Replace / AddObservation / AddMapPoint, and the isBad / IsInKeyFrame skips that depend on them)
 * are replayed in list order by the facade (facade/ORBmatcher.h), which owns the live map objects.
 * skip[i] != 0: NULL entry.  check_reprojection = 1 for the first overload (mono chi2 gate 5.99, :1138-1145), 0 for the Sim3
 * overload (Tcw7 = SE3f(Scw.rotationMatrix(), Scw.translation() / Scw.scale()), Ow3 = Tcw.inverse().translation()). */
int rumi_fuse_candidates(RumiMatcher *m, const RumiFrameFeatures *KF, float log_scale_factor, const float *Tcw7, const float *Ow3,
                         const float *K4, int32_t nmp, const uint8_t *skip, const float *mp_pos, const float *mp_normal,
                         const float *mp_min_dist, const float *mp_max_dist, const uint8_t *mp_desc, float th,
                         int32_t check_reprojection, int32_t *best_idx);

/* ORBmatcher::SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12, const Sim3f &S12, th) —
 * R/lib_src/ORBmatcher.cc:1293-1496 (loop / merge Sim3 refinement).  One entry per key-frame feature on each side:
 * skip1[i] = no map point, already matched (vbAlreadyMatched1) or bad; pc1_in2[i] = S21 * (T1w * p3Dw) — the point in camera 2,
 * formed by the caller with the reference's own Sophus expressions (:1338-1340); min/max_dist = mfMinDistance / mfMaxDistance;
 * desc = GetDescriptor().  Likewise side 2 with pc2_in1 = S12 * (T2w * p3Dw) and vbAlreadyMatched2.  K4 = pKF1's intrinsics
 * (the reference projects both directions with them, :1294-1297).  match12[i1] = KF2 feature whose map point agrees both
 * ways, or -1 (then vpMatches12[i1] is left as it was). */
int rumi_search_by_sim3(RumiMatcher *m, const RumiFrameFeatures *KF1, const RumiFrameFeatures *KF2, const float *K4,
                        float log_scale_factor, const uint8_t *skip1, const float *pc1_in2, const float *min_dist1,
                        const float *max_dist1, const uint8_t *desc1, const uint8_t *skip2, const float *pc2_in1, const float *min_dist2,
                        const float *max_dist2, const uint8_t *desc2, float th, int32_t *match12, int32_t *nfound_out);

/* Frame::isInFrustum(MapPoint*, viewingCosLimit) for every local map point (SearchLocalPoints, Tracking.cc:2996-3055;
 * Frame.cc:558-617, mono branch) — the step that produces the per-point inputs of rumi_search_by_projection_mappoints.
 * Rcw9 (row-major) = Frame::mRcw, tcw3 = mtcw, Ow3 = mOw; per point GetWorldPos, GetNormal, mfMinDistance, mfMaxDistance.
 * Outputs = the fields the reference writes on the MapPoint: mbTrackInView, mTrackProjX/Y (-1 when not projected inside the
 * image), mnTrackScaleLevel, mTrackViewCos, mTrackDepth. */
int rumi_frame_is_in_frustum(RumiMatcher *m, const float *Rcw9, const float *tcw3, const float *Ow3, const float *K4, float min_x,
                             float min_y, float max_x, float max_y, float log_scale_factor, int32_t nlevels, float viewing_cos_limit,
                             int32_t nmp, const float *mp_pos, const float *mp_normal, const float *mp_min_dist,
                             const float *mp_max_dist, uint8_t *track_in_view, float *proj_x, float *proj_y, int32_t *scale_level,
                             float *view_cos, float *track_depth);

/* Tracking::SearchLocalPoints, second half (R/lib_src/Tracking.cc:3012-3054): Frame::isInFrustum(pMP, 0.5) for every local map point and
 * ORBmatcher(0.8).SearchByProjection(mCurrentFrame, mvpLocalMapPoints, th, bFarPoints, thFarPoints) in ONE call — the frustum test's
 * per-point fields stay on the device and feed the search directly (no host round trip between the two).
 * skip[i] != 0: the point is not evaluated (mnLastFrameSeen == frame id, or isBad(); :3018-3021).  Pose and intrinsics as for
 * rumi_frame_is_in_frustum (image bounds from F), map-point fields as for rumi_frame_is_in_frustum + GetDescriptor() / Observations().
 * Outputs: the six tracking fields per point (the facade writes them on the non-skipped points and calls IncreaseVisible on those in
 * view), *n_to_match_out (nToMatch), frame_mp in/out and *nmatches_out as for rumi_search_by_projection_mappoints. */
int rumi_search_local_points(RumiMatcher *m, const RumiFrameFeatures *F, const float *Rcw9, const float *tcw3, const float *Ow3,
                             const float *K4, float log_scale_factor, int32_t nlevels, float viewing_cos_limit, int32_t nmp, const uint8_t *skip,
                             const float *mp_pos, const float *mp_normal, const float *mp_min_dist, const float *mp_max_dist,
                             const uint8_t *mp_desc, const int32_t *mp_obs, float th, int32_t far_points, float th_far_points, float nnratio,
                             uint8_t *track_in_view, float *proj_x, float *proj_y, int32_t *scale_level, float *view_cos, float *track_depth,
                             int32_t *n_to_match_out, int32_t *frame_mp, int32_t *nmatches_out);

/* Brute-force all-pairs 256-bit Hamming (the GPU formulation of BASELINE.json config 3), device pointers:
 * for each of B frame pairs, every query descriptor against every train descriptor; best index (first train index
 * wins ties, as in every loop of the reference), best and second-best distance.
 * d_query/d_train: [B][cap][32] u8; d_nq/d_nt: [B] int32 at stride `count_stride` int32s (pass the extractor's
 * [B][2] counts with count_stride = 2); outputs [B][cap] int32 each. */
int rumi_match_bruteforce_batch_device(const void *d_query, const void *d_nq, const void *d_train, const void *d_nt,
                                       int32_t count_stride, int32_t cap, int32_t nbatch, void *d_best_idx,
                                       void *d_best_dist, void *d_second_dist, void *hip_stream);
/* The same for descriptor blocks that are not densely packed: frame b's descriptors start at d_query + b * query_stride bytes (likewise
 * train); e.g. the per-frame records of rumi_orb_extract_batch_records_async (stride = record size, counts at count_stride = record size / 4). */
int rumi_match_bruteforce_batch_device_strided(const void *d_query, const void *d_nq, const void *d_train, const void *d_nt,
                                               int32_t count_stride, int64_t query_stride, int64_t train_stride, int32_t cap, int32_t nbatch,
                                               void *d_best_idx, void *d_best_dist, void *d_second_dist, void *hip_stream);
/* The consecutive-frame matching of a rumination queue (BASELINE.json configs[2]/[4]) in ONE launch: frame i of the buffer against frame i + 1,
 * the last one against the first.  d_desc: nframes descriptor blocks `frame_stride` bytes apart, d_n: their counts `count_stride` int32s apart;
 * outputs [nframes][cap] int32 each (row i = the matches of frame i in its successor). */
int rumi_match_bruteforce_ring_device(const void *d_desc, const void *d_n, int32_t count_stride, int64_t frame_stride, int32_t cap, int32_t nframes,
                                      void *d_best_idx, void *d_best_dist, void *d_second_dist, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* RUMI_MATCH_H */
