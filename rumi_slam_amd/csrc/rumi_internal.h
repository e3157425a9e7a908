// Entry points one translation unit of librumi_hip.so offers another (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace rumi {

// Correspondences the LDS instantiation of k_pose_opt holds (opt.hip).  The tracker (match.hip) launches ONLY that instantiation when it knows a
// frame cannot have more (nfeatures 1000 + the extractor's slack of 96 fits), so both files take the number from here.
constexpr int kPoseLdsEdges = 1152;

// Optimizer::PoseOptimization of ONE frame whose correspondences already lie on the device (opt.hip, k_pose_opt): dStart = {0, n} (n read by the
// kernel, not by the host), Xw [n][3], obs [n][2], w [n] (invLevelSigma2), K4, Tin [7] -> Tout [7], outlier [n], nGood [1].  dActive [cap] and
// dLastChi2 [cap] are scratch for frames of more than 1024 correspondences; fitsLds: the caller knows the count is at most 1024 (the second
// instantiation is then not launched).  Enqueues on `st`, does not synchronise.
int pose_opt_device(const int32_t *dStart, const float *dXw, const float *dObs, const float *dW, const float *dK4, const float *dTin, float *dTout,
                    uint8_t *dOutlier, int32_t *dNGood, uint8_t *dActive, double *dLastChi2, bool fitsLds, hipStream_t st);

}  // namespace rumi
