/*
 * rumi_opt.h — C ABI of the MI355X-native pose optimisation and local bundle adjustment (librumi_hip.so).
 *
 * Drop-in boundary for the two hot members of the all-static class ORB_SLAM3::Optimizer
 * (R/ = /root/reference/src/rumi-slam/, G/ = R/Thirdparty/g2o/g2o/):
 *   static int  PoseOptimization(Frame *pFrame)                                  R/include/cloud_edge_slam_lib/Optimizer.h:55, R/lib_src/Optimizer.cc:723-1001
 *   static void LocalBundleAdjustment(KeyFrame*, bool *pbStopFlag, Map*, int&x4) R/include/cloud_edge_slam_lib/Optimizer.h:53, R/lib_src/Optimizer.cc:1003-1355
 * and the g2o Levenberg-Marquardt / Huber / Schur machinery they run on (G/core/optimization_algorithm_levenberg.cpp:61-194,
 * G/core/block_solver.hpp:353-604, G/core/base_{unary,binary}_edge.hpp, G/core/robust_kernel_impl.cpp:78-91, G/types/se3quat.h).
 * The facade gathers the graph exactly as Optimizer.cc:763-897 / :1011-1271 do (under the same mutexes) and hands it over flat;
 * erasing observations and writing poses / points back (Optimizer.cc:1325-1354) stays in the facade under Map::mMutexMapUpdate.
 * Arithmetic is double precision on the device, float at both ends, as in the reference.  Results agree with the reference
 * algorithm to 1e-4 relative (the reference's own edge order is pointer-order dependent).
 *
 * Status codes, error string and threading rules: rumi_orb.h.
 */
#ifndef RUMI_OPT_H
#define RUMI_OPT_H

#include <stdint.h>

#include "rumi_orb.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct RumiOptimizer RumiOptimizer;

/* Scratch arenas: pose problems of up to max_pose_edges correspondences in total per call (all problems of a batch),
 * BA problems of up to max_kf key-frame vertices, max_mp points, max_edges observations. */
int rumi_opt_create(int32_t max_pose_edges, int32_t max_pose_batch, int32_t max_kf, int32_t max_mp, int32_t max_edges,
                    int32_t device, RumiOptimizer **out);
void rumi_opt_destroy(RumiOptimizer *o);

/* Optimizer::PoseOptimization.  One entry per feature of the frame that holds a map point, in feature order:
 * Xw n x 3 (MapPoint::GetWorldPos), obs n x 2 (mvKeysUn[i].pt), inv_sigma2 n (mvInvLevelSigma2[octave]); K4 = fx,fy,cx,cy;
 * Tcw7 in/out = pFrame->GetPose() / SetPose() as (qx,qy,qz,qw,tx,ty,tz); outlier_out n (mvbOutlier).
 * *n_good_out = the reference's return value nInitialCorrespondences - nBad (0, pose untouched, when n < 3). */
int rumi_pose_optimization(RumiOptimizer *o, const float *Xw, const float *obs, const float *inv_sigma2, int32_t n,
                           const float *K4, float *Tcw7, uint8_t *outlier_out, int32_t *n_good_out);

/* The same for B independent frames in one launch (one workgroup per frame).  Problem b owns entries
 * [start[b], start[b+1]) of Xw / obs / inv_sigma2 / outlier_out; Tcw7 is B x 7, n_good_out is B. */
int rumi_pose_optimization_batch(RumiOptimizer *o, int32_t nbatch, const int32_t *start, const float *Xw, const float *obs,
                                 const float *inv_sigma2, const float *K4, float *Tcw7, uint8_t *outlier_out,
                                 int32_t *n_good_out);

/* Optimizer::LocalBundleAdjustment on the flattened graph:
 *   kf_pose7  nKF x 7 in/out  key-frame poses Tcw (local key-frames first or in any order); fixed ones are not written
 *   kf_fixed  nKF             1 = fixed vertex (lFixedCameras, or the map's initial key-frame)
 *   mp_pos3   nMP x 3 in/out  map point positions
 *   edges     nE: e_mp / e_kf indices, e_obs nE x 2 (mvKeysUn pt), e_inv_sigma2 nE     (grouped by map point, as the
 *             reference builds them; any order is accepted)
 *   stop_flag the reference's pbStopFlag (host memory written by another thread; polled between LM trials; may be NULL)
 *   erase_out nE: 1 where the reference would erase the observation (chi2 > 5.991 or depth <= 0, Optimizer.cc:1285-1297)
 *   stats[4]: LM iterations run, LM trials run, number of optimised key-frames, 1 if aborted by the stop flag before start.
 * Returns RUMI_E_INVALID when there is no fixed key-frame (the reference returns silently, Optimizer.cc:1057-1060). */
int rumi_local_ba(RumiOptimizer *o, int32_t nKF, float *kf_pose7, const uint8_t *kf_fixed, int32_t nMP, float *mp_pos3,
                  int32_t nE, const int32_t *e_mp, const int32_t *e_kf, const float *e_obs, const float *e_inv_sigma2,
                  const float *K4, const volatile uint8_t *stop_flag, uint8_t *erase_out, int32_t *stats);

/* R independent windows of Optimizer::LocalBundleAdjustment (Optimizer.cc:1003-1355) in one call.  A window does not shard over GPUs or
 * workgroups ("replicas only"), but windows are independent: n_workers host threads, each with its own child handle
 * (stream + arenas, created on the first call and kept by `o`), take the windows from a shared counter, so the kernels of one window run while
 * another window's thread waits for the scalars of its LM trial.  Per window: the arguments of rumi_local_ba, stats[4] and the status of its
 * own run.  Returns the worst status. */
typedef struct RumiBaWindow {
    int32_t n_kf; float *kf_pose7; const uint8_t *kf_fixed;
    int32_t n_mp; float *mp_pos3;
    int32_t n_edges; const int32_t *e_mp, *e_kf; const float *e_obs, *e_inv_sigma2;
    const float *K4;
    const volatile uint8_t *stop_flag;
    uint8_t *erase_out;
    int32_t stats[4];
    int32_t status;
} RumiBaWindow;
int rumi_local_ba_batch(RumiOptimizer *o, int32_t n_windows, RumiBaWindow *windows, int32_t n_workers);

/* Optimizer::LocalBundleAdjustment(KeyFrame *pMainKF, vector<KeyFrame*> vpAdjustKF, vector<KeyFrame*> vpFixedKF, bool *pbStopFlag)
 * — the merge / welding-window bundle adjustment, R/lib_src/Optimizer.cc:3768-4183 (LoopClosing::MergeLocal, CloudMerging),
 * monocular edges.  Same flattened graph and the same outputs as rumi_local_ba; what differs is the procedure: optimize(5) with
 * Huber(sqrt(5.99)), then — unless *stop_flag — edges with chi2 > 5.991 or non-positive depth leave the optimisation (level 1),
 * the robust kernels are removed and optimize(10) runs again (:3986-4031); erase_out is the final test of :4042-4056.  A graph
 * without fixed key-frames is accepted, as upstream.  stats = {iterations of the first optimize, LM trials in total, optimised
 * key-frames, iterations of the second optimize}. */
int rumi_merge_ba(RumiOptimizer *o, int32_t nKF, float *kf_pose7, const uint8_t *kf_fixed, int32_t nMP, float *mp_pos3,
                  int32_t nE, const int32_t *e_mp, const int32_t *e_kf, const float *e_obs, const float *e_inv_sigma2,
                  const float *K4, const volatile uint8_t *stop_flag, uint8_t *erase_out, int32_t *stats);

/* Device time of the last rumi_local_ba call by stage, in ms (HIP events):
 * [0] linearise+Hll/Hpl, [1] pose block J^T W J on f64 MFMA, [2] Schur complement, [3] reduced solve, [4] update+chi2, [5] total */
int rumi_opt_stage_ms(RumiOptimizer *o, float ms[8]);
/* Opt-in per-kernel timing of the bundle adjustment: with profiling on, the next rumi_local_ba / rumi_merge_ba / rumi_bundle_adjustment call on
 * the dense-panel path records HIP events around the pose-block Gram product (f64 MFMA), the Schur-complement SYRK (f64 MFMA) and the
 * reduced solve of every LM trial; rumi_opt_kernel_ms returns their sums in ms and the trial count: [0] hpp, [1] syrk, [2] solve, [3] trials. */
int rumi_opt_set_profiling(RumiOptimizer *o, int32_t on);
int rumi_opt_kernel_ms(RumiOptimizer *o, float ms[4]);

/* Optimizer::BundleAdjustment(vpKFs, vpMP, nIterations, pbStopFlag, nLoopKF, bRobust) — R/lib_src/Optimizer.cc:54-351, monocular edges: the
 * full BA behind GlobalBundleAdjustemnt (map initialisation, Tracking.cc CreateInitialMapMonocular: 2 key-frames, 20 iterations; loop /
 * merge correction).  Same flattened graph as rumi_local_ba (kf_fixed[k] = the key-frame is the map's first one, :116); one
 * optimize(n_iterations) with Huber(sqrt(5.99)) when robust.  Landmarks without an edge keep their position (:248-250).  As for the
 * other BA entry points, up to 42 optimised key-frames take the dense-panel path; larger windows (up to 2560 optimised key-frames, within
 * the handle's max_kf) accumulate the Schur complement block-sparsely and factor it with a multi-workgroup blocked Cholesky.
 * stats = {LM iterations, LM trials, optimised key-frames, 0}. */
int rumi_bundle_adjustment(RumiOptimizer *o, int32_t nKF, float *kf_pose7, const uint8_t *kf_fixed, int32_t nMP, float *mp_pos3, int32_t nE,
                           const int32_t *e_mp, const int32_t *e_kf, const float *e_obs, const float *e_inv_sigma2, const float *K4,
                           const volatile uint8_t *stop_flag, int32_t n_iterations, int32_t robust, int32_t *stats);

/* Sim3Solver::ComputeInliersNum(map1KFs, map2KFs, avpValidKPMatches, gSw1w2) — R/lib_src/Sim3Solver.cc:564-664, the alignment score of
 * the rumination sub-map merge (CloudMerging.cc:611,748,809).  One entry per matched key-point pair, concatenated over the key-frame
 * pairs (pair_start [n_pairs + 1]); per pair the two composed transforms gSc1w2 = gSc1w1 * gSw1w2 and gSc2w1 = gSc2w2 * gSw1w2^-1 as
 * (qx qy qz qw tx ty tz s) doubles, formed by the caller with g2o::Sim3 itself (:620-621); per match the two map points' world
 * positions, the two key-points (mvKeys), mvLevelSigma2 at their octaves and the points' isEdge flags.  pair_denominator[p] =
 * vpValidKPMatches.size() (matches skipped for NULL map points still count, :644).  Outputs: inlier flag per match, ratio per pair,
 * and the returned value: the sorted ratios' element [n/2]. */
int rumi_sim3_inliers(RumiOptimizer *o, int32_t n_pairs, const int32_t *pair_start, const int32_t *pair_denominator, const double *S_c1w2,
                      const double *S_c2w1, const float *K4_1, const float *K4_2, const float *X1, const float *X2, const float *kp1,
                      const float *kp2, const float *sigma2_1, const float *sigma2_2, const uint8_t *edge1, const uint8_t *edge2,
                      uint8_t *inlier_out, float *ratio_out, float *median_out);

/* The alignment-score data of Sim3Solver::ComputeInliersNum (see rumi_sim3_inliers) for the hypotheses of rumi_sim3_ransac: instead of the
 * composed transforms the caller passes g2o::Sim3(R, t, 1.0) of the two key-frames of every pair (Sim3Solver.cc:596-597) and of the solver's
 * own key-frames mpKF1 / mpKF2 (:342-343), as (qx qy qz qw tx ty tz s) doubles; the composition with every hypothesis happens on the device. */
typedef struct RumiSim3ScoreSet {
    int32_t n_pairs;
    const int32_t *pair_start, *pair_denominator;
    const double *S_c1w1, *S_c2w2;
    const double *S_kf1w, *S_kf2w;
    const float *K4_1, *K4_2, *X1, *X2, *kp1, *kp2, *sigma2_1, *sigma2_2;
    const uint8_t *edge1, *edge2;
} RumiSim3ScoreSet;

/* Sim3Solver::iterate — R/lib_src/Sim3Solver.cc:159-220, :222-290 and the rumination overload :292-404 (CloudMerging.cc:717): the hypotheses of one
 * block of RANSAC iterations evaluated together, one workgroup each.  The n correspondences are those the constructor keeps (:78-126): both points in
 * their own camera frames (mvX3Dc1 / mvX3Dc2) and mvLevelSigma2 at the two key-points' octaves (mvnMaxError = 9.210 * sigma2, truncated to
 * size_t as upstream's vector<size_t> does; mvP1im1 / mvP2im2 are re-projected on the device, :128-129).  triples[3h..3h+2] = the minimal set of
 * hypothesis h, drawn by the caller exactly as :181-191 (DUtils::Random::RandomInt over the shrinking vAvailableIndices) — the draws do not depend on
 * earlier results, so a block of iterations is data-parallel and the caller replays the sequential "best so far / converged" logic of :198-214,
 * :277-283 or :349-373 over the per-hypothesis outputs (rumi_slam_amd.sim3solver.Sim3Solver, facade/Sim3Solver.h).
 * Per hypothesis: ComputeSim3 (:437-540, Horn 1987) and CheckInliers (:542-562); with score != NULL also ComputeInliersNum (:564-664) under
 * gSw1w2 = gSc1w^-1 * gSc1c2 * gSc2w (:338-347).  T12_out[h] = {mR12i row-major (9), mt12i (3), ms12i, valid, 0, 0}: valid = 0 marks the
 * degenerate set on which upstream returns early and keeps the previous iteration's transform (:493-494); n_inliers_out[h] = mnInliersi;
 * inlier_out [n_hyp][n] (may be NULL) = mvbInliersi; ratio_out [n_hyp][n_pairs] (may be NULL), median_out [n_hyp] = the returned ratio.
 * Upstream solves the 4x4 eigen-problem with Eigen::EigenSolver<Matrix4f> (not in the tree); here: Jacobi rotations in double on the same
 * float matrix — the transforms agree to float rounding, not bit for bit ("parity unpinned"). */
int rumi_sim3_ransac(RumiOptimizer *o, int32_t n, const float *X3Dc1, const float *X3Dc2, const float *sigma2_1, const float *sigma2_2,
                     const float *K4_1, const float *K4_2, int32_t fix_scale, int32_t n_hyp, const int32_t *triples, const RumiSim3ScoreSet *score,
                     float *T12_out, int32_t *n_inliers_out, uint8_t *inlier_out, float *ratio_out, float *median_out);

/* Optimizer::OptimizeSim3(pKF1, pKF2, vpMatches1, g2oS12, th2, bFixScale, mAcumHessian, bAllPoints) — R/lib_src/Optimizer.cc:1920-2167
 * (LoopClosing.cc:522,728, CloudMerging.cc:965,1169) and Optimizer::OptimizeCloudSim3(map1KFs, map2KFs, avpMatches, gSw1w2, th2, bFixScale,
 * mAcumHessian, bAllPoints) — :2169-2471 (CloudMerging.cc:803): Levenberg-Marquardt over ONE Sim3 vertex with numeric Jacobians,
 * optimize(5), removal of the correspondences whose edges exceed th2, optimize(10 or 5) without robust kernel, final inlier count.
 * One entry per correspondence that reaches "nCorrespondences++" (:2035 / :2316; the caller applies the map-point / isBad / i2 /
 * depth filters before): both points in their own camera frames (P3D1c, P3D2c as float, :1986-1998), the two observations (obs1 =
 * mvKeysUn of key-frame 1; obs2 = mvKeysUn of key-frame 2 or the float-normalised projection of :2065-2071, :2354-2360), mvInvLevelSigma2 at the two
 * octaves, skip12 / skip21 = the edge is not created (pMP2->isEdge / pMP1->isEdge, :2323,:2366; NULL = none).
 * S_c1w == NULL selects EdgeSim3ProjectXYZ / EdgeInverseSim3ProjectXYZ (OptimizeSim3: the vertex maps camera 2 into camera 1); otherwise
 * the world edges of OptimizeCloudSim3 with per key-frame-pair g2o::Sim3(R, t, 1.0) of both key-frames (:2231-2232) as
 * (qx qy qz qw tx ty tz s) doubles formed by the caller, and pair_of[i] = key-frame pair of correspondence i.
 * robust_first_pass: Huber(sqrt(th2)) on the first optimize (OptimizeSim3 :2050-2052; OptimizeCloudSim3 passes none, :2335,:2378).
 * S_io8: vertex estimate in / out (on the early return it holds the estimate after the first optimize, which OptimizeCloudSim3 has
 * already published at :2397 and OptimizeSim3 discards).  status_out[i]: 0 inlier, 1 removed after the first optimize, 2 fails the final
 * test, 3 kept but uncounted (one of its edges absent).  result3 = {nIn, nBad, 1 if "nCorrespondences - nBad < 10" returned early}.
 * mAcumHessian is zeroed by the reference and never accumulated (:2144, :2446): the facade does the same. */
int rumi_optimize_sim3(RumiOptimizer *o, int32_t n, const int32_t *pair_of, int32_t n_pairs, const double *S_c1w, const double *S_c2w,
                       const float *P1c, const float *P2c, const float *obs1, const float *obs2, const float *inv_sigma2_1,
                       const float *inv_sigma2_2, const uint8_t *skip12, const uint8_t *skip21, const float *K4_1, const float *K4_2, float th2,
                       int32_t fix_scale, int32_t robust_first_pass, double *S_io8, uint8_t *status_out, int32_t *result3);

#ifdef __cplusplus
}
#endif
#endif /* RUMI_OPT_H */
