// Definitions of the hot members of the reference's all-static ORB_SLAM3::Optimizer, with the reference's exact signatures
// (include/cloud_edge_slam_lib/Optimizer.h:45-92), each a forward to the MI355X facade templates (../Optimizer.h).  The reference's
// Optimizer.cc keeps every other member (inertial, essential graph, ...): build this file NEXT TO lib_src/Optimizer.cc and remove (or put
// under #ifndef RUMI_HIP) the reference's own definitions of
//   BundleAdjustment (:54-351), GlobalBundleAdjustemnt (:48-52), PoseOptimization (:723-1001), LocalBundleAdjustment (:1003-1355),
//   OptimizeSim3 (:1920-2167), OptimizeCloudSim3 (:2169-2471), LocalBundleAdjustment for the welding window (:3768-4183).
#include "Optimizer.h"           // the REFERENCE's header

#include "Frame.h"
#include "KeyFrame.h"
#include "Map.h"
#include "MapPoint.h"

#define RUMI_FACADE_NAMESPACE rumi_facade_impl
#include "../Optimizer.h"        // the facade templates (this repository)

namespace ORB_SLAM3 {

void Optimizer::BundleAdjustment(const std::vector<KeyFrame *> &vpKF, const std::vector<MapPoint *> &vpMP, int nIterations, bool *pbStopFlag,
                                 const unsigned long nLoopKF, const bool bRobust) {
    rumi_facade_impl::Optimizer::BundleAdjustment(vpKF, vpMP, nIterations, pbStopFlag, nLoopKF, bRobust);
}
void Optimizer::GlobalBundleAdjustemnt(Map *pMap, int nIterations, bool *pbStopFlag, const unsigned long nLoopKF, const bool bRobust) {
    rumi_facade_impl::Optimizer::GlobalBundleAdjustemnt(pMap, nIterations, pbStopFlag, nLoopKF, bRobust);
}
void Optimizer::LocalBundleAdjustment(KeyFrame *pKF, bool *pbStopFlag, Map *pMap, int &num_fixedKF, int &num_OptKF, int &num_MPs, int &num_edges) {
    rumi_facade_impl::Optimizer::LocalBundleAdjustment(pKF, pbStopFlag, pMap, num_fixedKF, num_OptKF, num_MPs, num_edges);
}
void Optimizer::LocalBundleAdjustment(KeyFrame *pMainKF, std::vector<KeyFrame *> vpAdjustKF, std::vector<KeyFrame *> vpFixedKF, bool *pbStopFlag) {
    rumi_facade_impl::Optimizer::LocalBundleAdjustment(pMainKF, vpAdjustKF, vpFixedKF, pbStopFlag);
}
int Optimizer::PoseOptimization(Frame *pFrame) { return rumi_facade_impl::Optimizer::PoseOptimization(pFrame); }
int Optimizer::OptimizeSim3(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches1, g2o::Sim3 &g2oS12, const float th2, const bool bFixScale,
                            Eigen::Matrix<double, 7, 7> &mAcumHessian, const bool bAllPoints) {
    return rumi_facade_impl::Optimizer::OptimizeSim3(pKF1, pKF2, vpMatches1, g2oS12, th2, bFixScale, mAcumHessian, bAllPoints);
}
float Optimizer::OptimizeCloudSim3(const std::vector<KeyFrame *> &map1KFs, const std::vector<KeyFrame *> &map2KFs, const std::vector<std::vector<MapPoint *>> &avpMatches,
                                   g2o::Sim3 &g2oS12, const float th2, const bool bFixScale, Eigen::Matrix<double, 7, 7> &mAcumHessian, const bool bAllPoints) {
    return rumi_facade_impl::Optimizer::OptimizeCloudSim3(map1KFs, map2KFs, avpMatches, g2oS12, th2, bFixScale, mAcumHessian, bAllPoints);
}

}  // namespace ORB_SLAM3
