// SE(3) / camera / robust-kernel scalar math shared by the optimiser kernels (double precision), restating
// G/types/se3quat.h (exp :217-257, map :213, operator* :104-110, normalizeRotation :280-285), Eigen's
// Quaternion <-> Matrix3 conversions, Pinhole::project / projectJac (R/lib_src/CameraModels/Pinhole.cpp:35-49,71-81)
// and RobustKernelHuber::robustify (G/core/robust_kernel_impl.cpp:78-91).
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define RUMI_HD __host__ __device__ inline
#else
#define RUMI_HD inline
#endif

namespace rumi {
#include "opt_math_body.inc"
}  // namespace rumi

// The same types and functions once more with floating-point contraction (a * b + c as ONE fma): namespace rumi::fused, used by
// PoseOptimization only.  The library is built with -ffp-contract=off because the extractor's steering math and the bundle adjustment
// follow the reference's unfused x86 arithmetic operation by operation; the pose optimiser's parity bar is a tolerance (1e-4,
// tests/test_optimizer_gpu.py), its device math already departs from libm by ~1e-16 (m_rcp / m_rsqrt above), and its cost is the number
// of f64 instructions one wave issues: fusing removes about a fifth of them.
#pragma clang fp contract(fast)
namespace rumi { namespace fused {
#include "opt_math_body.inc"
} }  // namespace rumi::fused
#pragma clang fp contract(off)
