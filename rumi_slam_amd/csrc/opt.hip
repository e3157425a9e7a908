// MI355X-native pose optimisation and local bundle adjustment behind include/rumi_opt.h (kernels + host side).
//
// PoseOptimization  one 256-thread workgroup per frame runs the whole 4-round / 10-iteration Levenberg-Marquardt loop
//                   in-kernel: residuals + 2x6 Jacobians per lane, 6x6 normal equations by wave shuffles + LDS, 6x6
//                   Cholesky, SE3 exp update, Huber, outlier re-classification (Optimizer.cc:909-991).  Frames batch.
// LocalBundleAdjustment  host-driven LM trials over device-resident double-precision state (stop flag polled per trial):
//   k_ba_build      one lane per observation: residual, Huber weight, 2x3 / 2x6 Jacobians; H_ll, b_l (f64 atomics),
//                   H_pl per edge, and the sqrt(w)-scaled rows [J_pose | r] of the pose panel, stored per key-frame
//   k_ba_hpp_mfma   the pose block H_pp = J_p^T W J_p and b_p as a dense Gram contraction on the f64 matrix cores
//                   (v_mfma_f64_16x16x4_f64, one workgroup per key-frame) — the only GEMM-shaped piece of local BA
//   k_ba_dinv/yfill one lane per landmark: D^-1 = (H_ll + lambda I)^-1 = L L^T, z = L^T b_l; one lane per edge: H_pl L into the dense panel Y
//   k_ba_syrk_mfma  Schur complement Y Y^T (and Y z) as a split-K SYRK on the f64 matrix cores (the window's Y is ~75 % dense)
//   k_ba_solve_tiles reduced pose system up to 175 unknowns: 16 x 16 tiles in LDS, panel factorisation in registers, MFMA trailing updates
//   k_ba_solve      the same for 176..255 unknowns (matrix in L2): blocked Cholesky (panel 8) with the right-hand
//                   side as an extra row, two barriers per panel, backward substitution on one wave
//   k_ba_update     landmark back-substitution, oplus on poses / points into the TRIAL state, x^T(lambda x + b)
//   k_ba_chi2       robustified chi2 of a state
// Accept / reject just swaps the current and trial state pointers (g2o's push / pop / discardTop).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <atomic>
#include <chrono>
#include <thread>
#include <sys/prctl.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "opt_math.h"
#include "rumi_internal.h"
#include "rumi_common.h"
#include "rumi_opt.h"

namespace rumi {

typedef double v4f64 __attribute__((ext_vector_type(4)));

// ---- exchanges between the lanes of a wave without the LDS crossbar ----
// The value of lane ^ D.  D = 8, 2, 1: DPP (row_ror:8, quad_perm); D = 4 has no DPP form on gfx9 and goes through ds_bpermute.
template <int D> __device__ __forceinline__ double lane_xor_f64(double v) {
    static_assert(D == 8 || D == 4 || D == 2 || D == 1, "row-local exchanges");
    if constexpr (D == 4) return __shfl_xor(v, 4);
    constexpr int ctl = D == 8 ? 0x128 : D == 2 ? 0x4E : 0xB1;             // row_ror:8 | quad_perm:[2,3,0,1] | quad_perm:[1,0,3,2]
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctl, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctl, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// One butterfly step over lane bit D = 32 or 16: a lane WITHOUT the bit gets lo(own) + lo(lane ^ D), a lane WITH it hi(own) + hi(lane ^ D).
// v_permlane32_swap / v_permlane16_swap (gfx950) trade the upper half (odd rows) of `lo` for the lower half (even rows) of `hi`: afterwards
// the two registers hold, in every lane, the kept value and the partner's -- two swaps and an add instead of four selects and two ds_bpermute.
template <int D> __device__ __forceinline__ double swap_add_f64(double lo, double hi) {
    static_assert(D == 32 || D == 16, "half-wave or row swap");
    unsigned a0 = (unsigned)__double2loint(lo), a1 = (unsigned)__double2hiint(lo), b0 = (unsigned)__double2loint(hi), b1 = (unsigned)__double2hiint(hi);
    if constexpr (D == 32) {
        const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false), r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        a0 = r0[0]; b0 = r0[1]; a1 = r1[0]; b1 = r1[1];
    } else {
        const auto r0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false), r1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
        a0 = r0[0]; b0 = r0[1]; a1 = r1[0]; b1 = r1[1];
    }
    return __hiloint2double((int)a1, (int)a0) + __hiloint2double((int)b1, (int)b0);
}
// sum over the wave, the same bits in every lane (the pairing of the xor butterfly: lane ^ 32, ^ 16, ... ^ 1)
__device__ __forceinline__ double wave_allreduce_f64(double s) {
    s = swap_add_f64<32>(s, s); s = swap_add_f64<16>(s, s);
    s += lane_xor_f64<8>(s); s += lane_xor_f64<4>(s); s += lane_xor_f64<2>(s); s += lane_xor_f64<1>(s);
    return s;
}

// ---- block reduction of NV doubles per thread (NW waves, 4 by default); every thread gets the total ----
// (FENCE_FIRST = false: the caller alternates between two `red` buffers from call to call, so no wave can still be reading the one written here)
template <int NV, int NW = 4, bool FENCE_FIRST = true> __device__ __forceinline__ void block_sum(double (&v)[NV], double *red /* [NW][NV] */) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
    for (int k = 0; k < NV; k++) v[k] = wave_allreduce_f64(v[k]);
    if (FENCE_FIRST) __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NV; k++) red[wave * NV + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; k++) {
        double t = red[k] + red[NV + k];
        if (NW >= 4) t += red[2 * NV + k] + red[3 * NV + k];
        if (NW == 8) t += (red[4 * NV + k] + red[5 * NV + k]) + (red[6 * NV + k] + red[7 * NV + k]);
        v[k] = t;
    }
}

// The same for up to 64 values per thread by a butterfly that halves the values a lane carries at every step (a lane ends with ONE
// value summed over its wave): PAD - 1 exchanges instead of 6 per value, then one LDS round for the four waves.  PAD = 32 or 64 slots
// (NV rounded up); red: (NW + 1) * PAD doubles.
template <int D, int N, int PAD> __device__ __forceinline__ void butterfly_stage(double (&w)[PAD], int lane) {   // lanes with bit D keep the upper N values, the others the lower N
    if constexpr (N >= 1) {
        if constexpr (D >= 16) {
#pragma unroll
            for (int i = 0; i < N; i++) w[i] = swap_add_f64<D>(w[i], w[i + N]);
        } else {
            const bool up = (lane & D) != 0;
#pragma unroll
            for (int i = 0; i < N; i++) {
                const double send = up ? w[i] : w[i + N], keep = up ? w[i + N] : w[i];
                w[i] = keep + lane_xor_f64<D>(send);
            }
        }
        if constexpr (D > 1) butterfly_stage<D / 2, N / 2, PAD>(w, lane);
    }
}
template <int NV, int NW = 4, bool FENCE_FIRST = true> __device__ __forceinline__ void block_sum_butterfly(double (&v)[NV], double *red) {
    static_assert(NV <= 64, "at most 64 values");
    constexpr int PAD = NV <= 32 ? 32 : 64;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double w[PAD];
#pragma unroll
    for (int k = 0; k < PAD; k++) w[k] = k < NV ? v[k] : 0.0;
    butterfly_stage<32, PAD / 2, PAD>(w, lane);
    int idx = lane;
    if (PAD == 32) { w[0] += lane_xor_f64<1>(w[0]); idx = lane >> 1; }     // 32 slots: lane pairs hold the same slot
    if (FENCE_FIRST) __syncthreads();
    if (PAD == 64 || (lane & 1) == 0) red[wave * PAD + idx] = w[0];
    __syncthreads();
    if (threadIdx.x < PAD) {
        double t = red[threadIdx.x] + red[PAD + threadIdx.x];
        if (NW >= 4) t += red[2 * PAD + threadIdx.x] + red[3 * PAD + threadIdx.x];
        if (NW == 8) t += (red[4 * PAD + threadIdx.x] + red[5 * PAD + threadIdx.x]) + (red[6 * PAD + threadIdx.x] + red[7 * PAD + threadIdx.x]);
        red[NW * PAD + threadIdx.x] = t;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; k++) v[k] = red[NW * PAD + k];
}


// ==================================================================================================================
// PoseOptimization
// ==================================================================================================================
struct PoseArgs {
    const int32_t *start;
    const float *Xw, *obs, *w, *K4;
    const float *Tin;     // initial poses [nbatch][7]
    float *Tout;          // optimised poses [nbatch][7] (left as Tin where the reference returns early)
    uint8_t *outlier;
    int32_t *nGood;
    uint8_t *active;      // scratch, one per correspondence
    double *lastChi2;     // scratch, one per correspondence
    int skipSmall;        // the global-memory instantiation leaves the frames the LDS instantiation solves
};

// (compiled with floating-point contraction, on rumi::fused's copy of the math: opt_math.h says why)
#pragma clang fp contract(fast)
// LDS = true (frames of up to kPoseLdsEdges correspondences): the edge data, the active flags and the last chi2 of every edge
// live in LDS for the whole solve, so none of the ~60 passes over the edges waits for global memory.
// (-DRUMI_POSE_STAMP, tools/build_stamp_lib.sh: every wave of frame 0 prints where its cycles went)
#ifdef RUMI_POSE_STAMP
#define POSE_STAMP_DECL long long stT = clock64(), stSerial = 0, stPass = 0, stRed = 0, stChi = 0, stOther = 0; int stN = 0, stTr = 0
#define POSE_STAMP(acc) do { const long long now_ = clock64(); acc += now_ - stT; stT = now_; } while (0)
#define POSE_COUNT(c) c++
#else
#define POSE_STAMP_DECL
#define POSE_STAMP(acc)
#define POSE_COUNT(c)
#endif
// (kPoseLdsEdges: rumi_internal.h, shared with the tracker)
template <bool LDS, int NT>
__global__ __launch_bounds__(NT) void k_pose_opt(PoseArgs A) {
    constexpr int NW = NT / 64;
    __shared__ double redBuf[2 * (NW + 1) * 32];                            // two reduction buffers used in turn: a reduction then needs no barrier before its first store
    int redFlip = 0;
    auto next_red = [&]() { redFlip ^= 1; return redBuf + redFlip * (NW + 1) * 32; };
    __shared__ float sXw[LDS ? 3 * kPoseLdsEdges : 1], sObs[LDS ? 2 * kPoseLdsEdges : 1], sW[LDS ? kPoseLdsEdges : 1];
    __shared__ double sChi[LDS ? kPoseLdsEdges : 1];
    __shared__ uint8_t sAct[LDS ? kPoseLdsEdges : 1];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int s0 = A.start[b], n = A.start[b + 1] - s0;
    if (LDS && n > kPoseLdsEdges) return;                                  // such frames are solved by the global-memory instantiation
    if (!LDS && n <= kPoseLdsEdges && A.skipSmall) return;
    const float *Xw = LDS ? sXw : A.Xw + (size_t)s0 * 3, *obs = LDS ? sObs : A.obs + (size_t)s0 * 2, *wgt = LDS ? sW : A.w + s0;
    uint8_t *outlier = A.outlier + s0, *active = LDS ? sAct : A.active + s0;
    double *lastChi2 = LDS ? sChi : A.lastChi2 + s0;
    if (LDS) {
        for (int i = tid; i < 3 * n; i += NT) sXw[i] = A.Xw[(size_t)s0 * 3 + i];
        for (int i = tid; i < 2 * n; i += NT) sObs[i] = A.obs[(size_t)s0 * 2 + i];
        for (int i = tid; i < n; i += NT) sW[i] = A.w[s0 + i];
    }
    for (int i = tid; i < n; i += NT) { outlier[i] = 0; active[i] = 1; }
    if (LDS) __syncthreads();
    if (n < 3) {                                                            // Optimizer.cc:899-900: returns 0, pose untouched
        if (tid == 0) A.nGood[b] = 0;
        if (tid < 7) A.Tout[(size_t)b * 7 + tid] = A.Tin[(size_t)b * 7 + tid];
        return;
    }
    const fused::DCam cam{A.K4[0], A.K4[1], A.K4[2], A.K4[3]};
    const double delta = (double)(float)sqrt(5.991), dsqr = delta * delta;  // const float deltaMono = sqrt(5.991)
    const fused::DSE3 T0 = fused::se3_from_float7(A.Tin + (size_t)b * 7);
    fused::DSE3 T = T0;
    bool robust = true;
    int nBadRound = 0;
    POSE_STAMP_DECL;

    // (the estimate maps a point by its rotation MATRIX, built once per pass -- 9 multiply-adds an edge instead of the quaternion form's two
    // cross products; the quaternion is normalised by every update)
    double Rm[3][3];
    auto set_pose = [&](const fused::DSE3 &P) { fused::quat_to_matrix(P.r, Rm); };
    // a thread's first kRegEdges edges (all of them up to 512 correspondences) stay in registers as doubles for the whole solve: no LDS read and
    // no float -> double conversion in the ~60 passes; further edges are read from the LDS (or global) arrays
    constexpr int kRegEdges = 2;
    double eX[kRegEdges], eY[kRegEdges], eZ[kRegEdges], eU[kRegEdges], eV[kRegEdges], eW[kRegEdges];
#pragma unroll
    for (int k = 0; k < kRegEdges; k++) {
        const int i = min(tid + k * NT, n - 1);                              // (n >= 3 here)
        eX[k] = (double)Xw[3 * i]; eY[k] = (double)Xw[3 * i + 1]; eZ[k] = (double)Xw[3 * i + 2];
        eU[k] = (double)obs[2 * i]; eV[k] = (double)obs[2 * i + 1]; eW[k] = (double)wgt[i];
    }
    uint8_t rAct[kRegEdges];                                                // ... and so do their active flag and last chi2 (only the owning thread reads them)
    double rChi[kRegEdges];
#pragma unroll
    for (int k = 0; k < kRegEdges; k++) { rAct[k] = 1; rChi[k] = 0; }
    auto for_edges = [&](auto &&body) {
#pragma unroll
        for (int k = 0; k < kRegEdges; k++) { const int i = tid + k * NT; if (i < n) body(i, eX[k], eY[k], eZ[k], eU[k], eV[k], eW[k], rAct[k], rChi[k]); }
        for (int i = tid + kRegEdges * NT; i < n; i += NT)
            body(i, (double)Xw[3 * i], (double)Xw[3 * i + 1], (double)Xw[3 * i + 2], (double)obs[2 * i], (double)obs[2 * i + 1], (double)wgt[i], active[i], lastChi2[i]);
    };
    auto edge_chi2 = [&](double X, double Y, double Z, double ou, double ov, double w, const fused::DSE3 &P, double &e0, double &e1, fused::D3 &pc) -> double {
        pc = fused::D3{Rm[0][0] * X + Rm[0][1] * Y + Rm[0][2] * Z + P.t.x, Rm[1][0] * X + Rm[1][1] * Y + Rm[1][2] * Z + P.t.y,
                       Rm[2][0] * X + Rm[2][1] * Y + Rm[2][2] * Z + P.t.z};
        double u, v;
        fused::cam_project(cam, pc, u, v);
        e0 = ou - u; e1 = ov - v;
        return e0 * w * e0 + e1 * w * e1;
    };
    auto robust_chi2 = [&](const fused::DSE3 &P) -> double {                      // computeActiveErrors + activeRobustChi2
        double acc[1] = {0};
        set_pose(P);
        for_edges([&](int i, double X, double Y, double Z, double ou, double ov, double w, uint8_t &act, double &last) {
            if (!act) return;
            double e0, e1; fused::D3 pc;
            const double c = edge_chi2(X, Y, Z, ou, ov, w, P, e0, e1, pc);
            last = c;
            double r0 = c, r1 = 1;
            if (robust) fused::huber(c, delta, dsqr, r0, r1);
            acc[0] += r0;
        });
        block_sum<1, NW, false>(acc, next_red());
        return acc[0];
    };

    for (int it = 0; it < 4; it++) {
        T = T0;                                                            // estimate reset every round (:910-911)
        // (the edges active in this round: all in the first, then those the last re-classification kept -- no pass to count them)
        if ((it == 0 ? n : n - nBadRound) > 0) {
            // ---- g2o optimize(10): optimization_algorithm_levenberg.cpp:61-169 ----
            double lambda = -1, ni = 2;
            int nBad = 0;
            for (int itl = 0; itl < 10; itl++) {
                // computeActiveErrors + activeRobustChi2 and buildSystem evaluate every edge at the same estimate: one pass, the
                // robust chi2 rides along as the 28th reduced value (same per-edge values, same reduction tree as robust_chi2)
#ifdef RUMI_POSE_STAMP
                asm volatile("" : "+v"(lambda), "+v"(T.r.x), "+v"(T.t.x), "+v"(nBad));      // the trial's decisions are taken before the stamp
#endif
                POSE_STAMP(stSerial);
                double hb[28];                                             // 21 upper entries of H, 6 of b, robust chi2
#pragma unroll
                for (int k = 0; k < 28; k++) hb[k] = 0;
                set_pose(T);
                for_edges([&](int i, double X, double Y, double Z, double ou, double ov, double w, uint8_t &act, double &last) {
                    if (!act) return;
                    double e0, e1; fused::D3 pc;
                    const double c = edge_chi2(X, Y, Z, ou, ov, w, T, e0, e1, pc);
                    last = c;
                    double r0 = c, r1 = 1;
                    if (robust) fused::huber(c, delta, dsqr, r0, r1);
                    hb[27] += r0;
                    double J0[6], J1[6];
                    fused::jac_pose(cam, pc, J0, J1);
                    const double rw = r1 * w;
                    // H += rw J^T J, b -= r1 w J^T e with the weights multiplied into one factor first (two multiply-adds an entry); J0[4] and
                    // J1[3] are zero by construction (jac_pose): their products are left out, H[3][4] stays 0
                    double A0[6], A1[6];
#pragma unroll
                    for (int a = 0; a < 6; a++) { A0[a] = rw * J0[a]; A1[a] = rw * J1[a]; }
                    const double we0 = rw * e0, we1 = rw * e1;
                    int p = 0;
#pragma unroll
                    for (int a = 0; a < 6; a++) {
#pragma unroll
                        for (int c2 = a; c2 < 6; c2++, p++) {
                            const bool z0 = a == 4 || c2 == 4, z1 = a == 3 || c2 == 3;
                            if (z0 && z1) continue;
                            hb[p] += z0 ? A1[a] * J1[c2] : z1 ? A0[a] * J0[c2] : A0[a] * J0[c2] + A1[a] * J1[c2];
                        }
                    }
#pragma unroll
                    for (int a = 0; a < 6; a++) hb[21 + a] -= a == 4 ? J1[a] * we1 : a == 3 ? J0[a] * we0 : J0[a] * we0 + J1[a] * we1;
                });
                POSE_STAMP(stPass);
                block_sum_butterfly<28, NW, false>(hb, next_red());
                POSE_STAMP(stRed); POSE_COUNT(stN);
                double currentChi = hb[27];
                const double iniChi = currentChi;
                if (itl == 0) {                                            // computeLambdaInit: tau * max |H_jj|
                    double m = 0;
                    int p = 0;
                    for (int a = 0; a < 6; a++) { m = fmax(fabs(hb[p]), m); p += 6 - a; }
                    lambda = 1e-5 * m; ni = 2; nBad = 0;
                }
                double rho = 0;
                int qmax = 0;
                do {
                    const fused::DSE3 saved = T;                                  // push()
                    double x[6];
                    const bool ok2 = fused::chol_solve_packed<6>(hb, lambda, hb + 21, x);   // setLambda + solve + restoreDiagonal
                    if (ok2) T = fused::se3_mul(fused::se3_exp_series(x), T);                    // oplusImpl: exp(update) * estimate
                    POSE_STAMP(stSerial); POSE_COUNT(stTr);
                    double tempChi = robust_chi2(T);
                    POSE_STAMP(stChi);
                    if (!ok2) tempChi = DBL_MAX;
                    rho = currentChi - tempChi;
                    double scale = 0;
                    if (ok2) for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + hb[21 + j]);
                    scale += 1e-3;
                    rho *= fused::m_rcp(scale);                             // (v_rcp_f64 + two Newton steps instead of the IEEE divide sequence)
                    if (rho > 0 && isfinite(tempChi)) {
                        const double tr = 2 * rho - 1;
                        double alpha = 1. - tr * tr * tr;          // pow(2 rho - 1, 3) (levenberg.cpp:124): the generic pow is ~150 instructions of this serial section
                        alpha = fmin(alpha, 2. / 3.);
                        lambda *= fmax(1. / 3., alpha);
                        ni = 2;
                        currentChi = tempChi;
                    } else {
                        lambda *= ni;
                        ni *= 2;
                        T = saved;                                         // pop()
                    }
                    qmax++;
                } while (rho < 0 && qmax < 10);
                if (qmax == 10 || rho == 0) break;                         // Terminate
                if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
                if (nBad >= 3) break;
            }
        }
        // re-classification (:916-939): former outliers get a fresh error, active edges keep the last computed one
        double bad[1] = {0};
        set_pose(T);
        for_edges([&](int i, double X, double Y, double Z, double ou, double ov, double w, uint8_t &act, double &last) {
            double e0, e1; fused::D3 pc;
            const float chi2 = (float)(!act ? edge_chi2(X, Y, Z, ou, ov, w, T, e0, e1, pc) : last);       // (an edge is inactive exactly when it is an outlier)
            if (chi2 > 5.991f) { outlier[i] = 1; act = 0; bad[0] += 1; }
            else { outlier[i] = 0; act = 1; }
        });
        block_sum<1, NW, false>(bad, next_red());
        nBadRound = (int)bad[0];
        if (it == 2) robust = false;                                       // setRobustKernel(0)
        if (n < 10) break;                                                 // optimizer.edges().size() < 10
    }
#ifdef RUMI_POSE_STAMP
    POSE_STAMP(stOther);
    if (b == 0 && (tid & 63) == 0) printf("pose stamp wave %d n %d builds %d trials %d: serial %lld  build passes %lld  butterfly %lld  chi2 passes+reduce %lld  other %lld\n", tid >> 6, n, stN, stTr, stSerial, stPass, stRed, stChi, stOther);
#endif
    if (tid == 0) {
        fused::se3_to_float7(T, A.Tout + (size_t)b * 7);
        A.nGood[b] = n - nBadRound;
    }
}

#pragma clang fp contract(off)

// ==================================================================================================================
// LocalBundleAdjustment
// ==================================================================================================================
struct BADev {
    int nKF, nMP, nE, nOpt, n;                 // n = 6 nOpt
    const int32_t *eMP, *eKF, *poseCol;        // poseCol[kf] = column block or -1 (fixed)
    const int32_t *ptStart, *ptEdge;           // edges grouped by landmark (CSR)
    const int32_t *rowSlot;                    // edge -> first of its two rows in the key-frame-ordered pose panel (-1 fixed)
    const int32_t *kfRowStart;                 // [nOpt + 1] row ranges of the panel
    const float *obs, *info;                   // the caller's single-precision measurements and weights as they came (widened where they are read)
    DCam cam;
    double delta, dsqr;
    double *Hll, *bl, *Hpl, *panel, *Hpp, *bp, *Dinv, *S, *bs, *x, *lastChi2;
    double *scal;                              // [0] chi2, [1] scale, [2] max diag (as bits), [3] ok flag
    const uint8_t *off;                        // edge at level 1 (excluded from the optimisation), merge BA second pass
    int robust;                                // Huber kernel on the edges (off in the merge BA second pass)
};

// Device-side control of g2o's Levenberg-Marquardt loop (optimization_algorithm_levenberg.cpp:61-169) for the local windows.  The host enqueues
// one SLOT per LM trial -- [build the system if the previous trial was accepted] -> reduced system -> solve -> update -> chi2 -> decide -- ahead
// of the device; k_lm_decide takes the accept / reject decision, updates lambda and the iteration bookkeeping here, and every kernel of a slot
// reads this block to know what to do (nothing once `done`; which of the two state buffers is the estimate; the damping).  No host round
// trip inside the loop: the host only watches `trialsDone` / `done` in the pinned mirror to keep one slot queued ahead, and forwards the
// reference's stop flag (pbStopFlag, plain host memory) into `stopWord`.
struct LmState {
    double lambda, ni, currentChi, iniChi;
    double *T[2], *X[2];                 // the two state buffers; `cur` is the estimate, cur ^ 1 the trial
    int32_t cur, needBuild, done, nBad, qmax, it, maxIt, trials, iters, nUnknowns;
};
struct LmMirror {                        // fine-grained pinned host memory, written by k_lm_decide / k_lm_begin
    volatile int32_t trialsDone, done, cur, iters, trials, pad;
    volatile unsigned long long seq;
};
__device__ __forceinline__ bool lm_skip(const LmState *S, bool buildOnly) { return S && (S->done || (buildOnly && !S->needBuild)); }

__device__ __forceinline__ DSE3 load_pose(const double *T, int k) {
    const double *p = T + (size_t)k * 8;
    return DSE3{{p[0], p[1], p[2], p[3]}, {p[4], p[5], p[6]}};
}
__device__ __forceinline__ void store_pose(double *T, int k, const DSE3 &P) {
    double *p = T + (size_t)k * 8;
    p[0] = P.r.x; p[1] = P.r.y; p[2] = P.r.z; p[3] = P.r.w; p[4] = P.t.x; p[5] = P.t.y; p[6] = P.t.z; p[7] = 0;
}

__global__ __launch_bounds__(256) void k_ba_chi2(BADev B, const double *T, const double *X, const LmState *S = nullptr, int trial = 0) {
    __shared__ double red[4];
    if (S) { if (S->done) return; T = S->T[S->cur ^ trial]; X = S->X[S->cur ^ trial]; }
    const int e = blockIdx.x * 256 + threadIdx.x;
    double acc[1] = {0};
    if (e < B.nE) {
        const int p = B.eMP[e];
        const D3 pc = se3_map(load_pose(T, B.eKF[e]), D3{X[3 * p], X[3 * p + 1], X[3 * p + 2]});
        double u, v;
        cam_project(B.cam, pc, u, v);
        const double e0 = (double)B.obs[2 * e] - u, e1 = (double)B.obs[2 * e + 1] - v, w = (double)B.info[e];
        const double c = e0 * w * e0 + e1 * w * e1;
        if (!B.off[e]) {                                                    // level-1 edges keep the error of their last active pass
            B.lastChi2[e] = c;
            double r0 = c, r1 = 1;
            if (B.robust) huber(c, B.delta, B.dsqr, r0, r1);
            acc[0] = r0;
        }
    }
    block_sum<1>(acc, red);
    if (threadIdx.x == 0) atomicAdd(&B.scal[0], acc[0]);
}

// Lanes walk the edges in landmark order (ptEdge): the contributions to H_ll and b_l of one landmark sit in consecutive lanes and are summed
// by a segmented wave reduction, so only the first lane of every run issues atomics (device-scope f64 atomics are served past the per-XCD
// L2s: 12 per edge cost 37 us per call at 44 k edges, a twelfth of that 12 us).
__device__ __forceinline__ void seg_reduce_atomic(double v, int p, bool head, double *dst) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double vo = __shfl_down(v, d);
        const int po = __shfl_down(p, d);
        if (lane + d < 64 && po == p) v += vo;
    }
    if (head && p >= 0) atomicAdd(dst, v);
}

__global__ __launch_bounds__(256) void k_ba_build(BADev B, const double *T, const double *X, const LmState *S = nullptr) {
    if (S) { if (lm_skip(S, true)) return; T = S->T[S->cur]; X = S->X[S->cur]; }
    const int t = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    const bool liveEdge = t < B.nE;
    const int e = liveEdge ? B.ptEdge[t] : 0;
    const int pKey = liveEdge ? B.eMP[e] : -1;
    const int pPrev = __shfl_up(pKey, 1);
    const bool head = lane == 0 || pPrev != pKey;            // first lane of a landmark's run inside this wave
    double hl[6] = {0, 0, 0, 0, 0, 0}, blv[3] = {0, 0, 0};  // this edge's A^T W A (upper triangle) and -A^T W e
    if (liveEdge) {
    const int p = pKey, kf = B.eKF[e];
    const DSE3 P = load_pose(T, kf);
    const D3 pc = se3_map(P, D3{X[3 * p], X[3 * p + 1], X[3 * p + 2]});
    double u, v;
    cam_project(B.cam, pc, u, v);
    const double e0 = (double)B.obs[2 * e] - u, e1 = (double)B.obs[2 * e + 1] - v, info = (double)B.info[e];
    const double c = e0 * info * e0 + e1 * info * e1;
    if (B.off[e]) {                                                         // inactive edge: contributes nothing to H, b, Y
        const int slot0 = B.rowSlot[e];
        if (slot0 >= 0) {
            double *hp = B.Hpl + (size_t)e * 18, *rp = B.panel + (size_t)slot0 * 8;
#pragma unroll
            for (int k = 0; k < 18; k++) hp[k] = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) rp[k] = 0;
        }
    } else {
    double r0 = c, r1 = 1;
    if (B.robust) huber(c, B.delta, B.dsqr, r0, r1);
    const double w = r1 * info;
    double J0[6], J1[6], R[3][3], A0[3], A1[3];
    jac_pose(B.cam, pc, J0, J1);
    quat_to_matrix(P.r, R);
    const double iz = 1.0 / pc.z, iz2 = iz * iz;
    const double j00 = B.cam.fx * iz, j02 = -B.cam.fx * pc.x * iz2, j11 = B.cam.fy * iz, j12 = -B.cam.fy * pc.y * iz2;
#pragma unroll
    for (int k = 0; k < 3; k++) { A0[k] = -(j00 * R[0][k] + j02 * R[2][k]); A1[k] = -(j11 * R[1][k] + j12 * R[2][k]); }   // -projectJac * R
    // landmark block and right-hand side: summed over the landmark's run below
    {
        int q = 0;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            blv[a] = -w * (A0[a] * e0 + A1[a] * e1);
#pragma unroll
            for (int c2 = a; c2 < 3; c2++) hl[q++] = w * (A0[a] * A0[c2] + A1[a] * A1[c2]);
        }
    }
    const int slot = B.rowSlot[e];
    if (slot >= 0) {
        double *hp = B.Hpl + (size_t)e * 18;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int c2 = 0; c2 < 3; c2++) hp[a * 3 + c2] = w * (J0[a] * A0[c2] + J1[a] * A1[c2]);
        const double sw = sqrt(w);
        double *r0p = B.panel + (size_t)slot * 8, *r1p = r0p + 8;
#pragma unroll
        for (int a = 0; a < 6; a++) { r0p[a] = sw * J0[a]; r1p[a] = sw * J1[a]; }
        r0p[6] = sw * e0; r0p[7] = 0; r1p[6] = sw * e1; r1p[7] = 0;
    }
    }   // active edge
    }   // live edge
    double *Hl = B.Hll + (size_t)max(pKey, 0) * 9, *bL = B.bl + (size_t)max(pKey, 0) * 3;
    seg_reduce_atomic(blv[0], pKey, head, &bL[0]); seg_reduce_atomic(blv[1], pKey, head, &bL[1]); seg_reduce_atomic(blv[2], pKey, head, &bL[2]);
    // upper triangle 00 01 02 11 12 22; the mirrored entries get the same sums
    {
        const int at[6] = {0, 1, 2, 4, 5, 8}, mir[6] = {-1, 3, 6, -1, 7, -1};
#pragma unroll
        for (int q = 0; q < 6; q++) {
            double v = hl[q];
            const int lanei = lane;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const double vo = __shfl_down(v, d);
                const int po = __shfl_down(pKey, d);
                if (lanei + d < 64 && po == pKey) v += vo;
            }
            if (head && pKey >= 0) { atomicAdd(&Hl[at[q]], v); if (mir[q] >= 0) atomicAdd(&Hl[mir[q]], v); }
        }
    }
}

// H_pp(kf) = sum over the key-frame's rows of row^T row on the f64 matrix cores; [0:6,0:6] is the 6x6 block, -[0:6,6] is b_p.
// Lane l feeds element (row l>>4, column l&15) of a 4-row chunk as BOTH operands (A = chunk^T, B = chunk).
// grid (nOpt, kHppSlices): every wave owns an interleaved subset of the 4-row chunks (4 loads in flight per wave); the
// slices of one key-frame are combined with f64 atomics into the zeroed H_pp / b_p.
constexpr int kHppSlices = 16;
__global__ __launch_bounds__(256) void k_ba_hpp_mfma(BADev B, const LmState *S = nullptr) {
    if (lm_skip(S, true)) return;
    const int kf = blockIdx.x, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r0 = B.kfRowStart[kf], r1 = B.kfRowStart[kf + 1];
    const int col = lane & 15, sub = lane >> 4;
    const int nw = kHppSlices * 4, w = blockIdx.y * 4 + wave;
    constexpr int U = 16;                                  // 4-row chunks in flight per wave: the loop is bound by memory latency, not by the 64-cycle MFMAs
    v4f64 acc = {0, 0, 0, 0};
    if (r1 > r0) {
        for (int r = r0 + w * 4; r < r1; r += nw * 4 * U) {
            double v[U];
#pragma unroll
            for (int u = 0; u < U; u++) {                  // clamped address, no branch around the load
                const int row = r + u * nw * 4 + sub;
                v[u] = B.panel[(size_t)min(row, r1 - 1) * 8 + (col & 7)];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int row = r + u * nw * 4 + sub;
                const double x = (row < r1 && col < 8) ? v[u] : 0.0;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc, 0, 0, 0);
            }
        }
    }
    // D[row = sub + 4*reg][col]: rows 0..7 live in reg 0 and reg 1
#pragma unroll
    for (int reg = 0; reg < 2; reg++) {
        const int a = sub + 4 * reg, c = col;
        const double g = acc[reg];
        if (g != 0.0) {
            if (a < 6 && c < 6) atomicAdd(&B.Hpp[(size_t)kf * 36 + a * 6 + c], g);
            if (a < 6 && c == 6) atomicAdd(&B.bp[(size_t)kf * 6 + a], -g);
        }
    }
}

// one launch instead of a memset per array
struct ZeroList { double *p[4]; int n[4]; };
__global__ void k_ba_zero(ZeroList Z, const LmState *S = nullptr) {
    if (lm_skip(S, true)) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int s = 0; s < 4; s++) if (i < Z.n[s]) Z.p[s][i] = 0.0;
}

__global__ void k_ba_maxdiag(BADev B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double m = 0;
    if (i < B.nOpt * 6) m = fabs(B.Hpp[(size_t)(i / 6) * 36 + (i % 6) * 7]);
    else if (i < B.nOpt * 6 + B.nMP * 3) { const int j = i - B.nOpt * 6; m = fabs(B.Hll[(size_t)(j / 3) * 9 + (j % 3) * 4]); }
    else return;
    atomicMax(reinterpret_cast<unsigned long long *>(&B.scal[2]), (unsigned long long)__double_as_longlong(m));   // m >= 0: bit order = value order
}

// Schur complement as a dense SYRK on the f64 matrix cores.  With D^-1 = L L^T (3x3 Cholesky per landmark) the update is
//   S = H_pp + lambda I - Y Y^T,   b_s = b_p - Y z,     Y(:, 3p..3p+2) = stack of H_pl(e) L over the landmark's edges,  z = L^T b_l,
// and Y is ~75 % dense for a covisibility window (every landmark is seen by most key-frames), so the block-sparse loops of
// g2o (block_solver.hpp:379-438) become one Gram product.  Yt is stored K-major: row k = 3p+c holds the NP-padded column
// of Y plus z in entry n, so the augmented Gram matrix G = Yt^T Yt carries Y Y^T in G[0:n,0:n] and Y z in G[0:n,n].
__global__ void k_ba_dinv(BADev B, double lambda, double *Yt, int NP, double *Lp) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= B.nMP) return;
    double D[9];
#pragma unroll
    for (int i = 0; i < 9; i++) D[i] = B.Hll[(size_t)p * 9 + i];
    D[0] += lambda; D[4] += lambda; D[8] += lambda;
    const double c00 = D[4] * D[8] - D[5] * D[7], c01 = D[5] * D[6] - D[3] * D[8], c02 = D[3] * D[7] - D[4] * D[6];
    const double id = 1.0 / (D[0] * c00 + D[1] * c01 + D[2] * c02);        // Eigen Matrix3d::inverse (cofactors)
    double I[9];
    I[0] = c00 * id; I[1] = (D[2] * D[7] - D[1] * D[8]) * id; I[2] = (D[1] * D[5] - D[2] * D[4]) * id;
    I[3] = c01 * id; I[4] = (D[0] * D[8] - D[2] * D[6]) * id; I[5] = (D[2] * D[3] - D[0] * D[5]) * id;
    I[6] = c02 * id; I[7] = (D[1] * D[6] - D[0] * D[7]) * id; I[8] = (D[0] * D[4] - D[1] * D[3]) * id;
#pragma unroll
    for (int i = 0; i < 9; i++) B.Dinv[(size_t)p * 9 + i] = I[i];
    const double l00 = sqrt(I[0]), l10 = I[3] / l00, l20 = I[6] / l00;
    const double l11 = sqrt(I[4] - l10 * l10), l21 = (I[7] - l20 * l10) / l11, l22 = sqrt(I[8] - l20 * l20 - l21 * l21);
    double *L = Lp + (size_t)p * 6;
    L[0] = l00; L[1] = l10; L[2] = l20; L[3] = l11; L[4] = l21; L[5] = l22;
    const double b0 = B.bl[3 * p], b1 = B.bl[3 * p + 1], b2 = B.bl[3 * p + 2];
    double *y0 = Yt + (size_t)(3 * p) * NP;
    y0[B.n] = l00 * b0 + l10 * b1 + l20 * b2; y0[NP + B.n] = l11 * b1 + l21 * b2; y0[2 * NP + B.n] = l22 * b2;    // z = L^T b_l
}

__global__ void k_ba_yfill(BADev B, double *Yt, int NP, const double *Lp) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B.nE) return;
    const int col = B.poseCol[B.eKF[e]];
    if (col < 0) return;
    const int p = B.eMP[e];
    const double *L = Lp + (size_t)p * 6, *h = B.Hpl + (size_t)e * 18;
    const double l00 = L[0], l10 = L[1], l20 = L[2], l11 = L[3], l21 = L[4], l22 = L[5];
    double *y0 = Yt + (size_t)(3 * p) * NP + col * 6, *y1 = y0 + NP, *y2 = y1 + NP;
#pragma unroll
    for (int a = 0; a < 6; a++) {
        const double h0 = h[a * 3], h1 = h[a * 3 + 1], h2 = h[a * 3 + 2];
        y0[a] = h0 * l00 + h1 * l10 + h2 * l20;
        y1[a] = h1 * l11 + h2 * l21;
        y2[a] = h2 * l22;
    }
}

// k_ba_dinv and k_ba_yfill in one launch (one round of kernel-launch and memory latency less per LM trial): threads [0, nMP) do the landmark
// part; threads [nMP, nMP + nE) fill the panel from the factor of their landmark's block, which each recomputes (45 flops) rather than read.
__device__ __forceinline__ void dinv_factor(const double *Hll, double lambda, double (&I)[9], double (&L)[6]) {
    double D[9];
#pragma unroll
    for (int i = 0; i < 9; i++) D[i] = Hll[i];
    D[0] += lambda; D[4] += lambda; D[8] += lambda;
    const double c00 = D[4] * D[8] - D[5] * D[7], c01 = D[5] * D[6] - D[3] * D[8], c02 = D[3] * D[7] - D[4] * D[6];
    const double id = 1.0 / (D[0] * c00 + D[1] * c01 + D[2] * c02);        // Eigen Matrix3d::inverse (cofactors)
    I[0] = c00 * id; I[1] = (D[2] * D[7] - D[1] * D[8]) * id; I[2] = (D[1] * D[5] - D[2] * D[4]) * id;
    I[3] = c01 * id; I[4] = (D[0] * D[8] - D[2] * D[6]) * id; I[5] = (D[2] * D[3] - D[0] * D[5]) * id;
    I[6] = c02 * id; I[7] = (D[1] * D[6] - D[0] * D[7]) * id; I[8] = (D[0] * D[4] - D[1] * D[3]) * id;
    const double l00 = sqrt(I[0]), l10 = I[3] / l00, l20 = I[6] / l00;
    const double l11 = sqrt(I[4] - l10 * l10), l21 = (I[7] - l20 * l10) / l11, l22 = sqrt(I[8] - l20 * l20 - l21 * l21);
    L[0] = l00; L[1] = l10; L[2] = l20; L[3] = l11; L[4] = l21; L[5] = l22;
}
__global__ __launch_bounds__(256) void k_ba_dinv_yfill(BADev B, double lambda, double *Yt, int NP, double *Lp, const LmState *S = nullptr) {
    if (S) { if (S->done) return; lambda = S->lambda; }
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < B.nMP) {
        const int p = t;
        double I[9], L[6];
        dinv_factor(B.Hll + (size_t)p * 9, lambda, I, L);
#pragma unroll
        for (int i = 0; i < 9; i++) B.Dinv[(size_t)p * 9 + i] = I[i];
#pragma unroll
        for (int i = 0; i < 6; i++) Lp[(size_t)p * 6 + i] = L[i];
        const double b0 = B.bl[3 * p], b1 = B.bl[3 * p + 1], b2 = B.bl[3 * p + 2];
        double *y0 = Yt + (size_t)(3 * p) * NP;
        y0[B.n] = L[0] * b0 + L[1] * b1 + L[2] * b2; y0[NP + B.n] = L[3] * b1 + L[4] * b2; y0[2 * NP + B.n] = L[5] * b2;    // z = L^T b_l
        return;
    }
    const int e = t - B.nMP;
    if (e >= B.nE) return;
    const int col = B.poseCol[B.eKF[e]];
    if (col < 0) return;
    const int p = B.eMP[e];
    double I[9], L[6];
    dinv_factor(B.Hll + (size_t)p * 9, lambda, I, L);
    const double *h = B.Hpl + (size_t)e * 18;
    double *y0 = Yt + (size_t)(3 * p) * NP + col * 6, *y1 = y0 + NP, *y2 = y1 + NP;
#pragma unroll
    for (int a = 0; a < 6; a++) {
        const double h0 = h[a * 3], h1 = h[a * 3 + 1], h2 = h[a * 3 + 2];
        y0[a] = h0 * L[0] + h1 * L[1] + h2 * L[2];
        y1[a] = h1 * L[3] + h2 * L[4];
        y2[a] = h2 * L[5];
    }
}

// G += Yt^T Yt over a K-slice; one wave per (upper 16x16 tile, slice); grid tiles x slices/4, 256 threads.  A wave's 36-odd MFMAs take 2.3 k
// cycles, one round trip to L2 / HBM about as long: the slice is walked in chunks of 64 rows (16 operand pairs per lane) with the loads of two
// chunks in flight before the first MFMA, so the kernel pays the memory latency once per wave, not once per 16 rows.
__global__ __launch_bounds__(256) void k_ba_syrk_mfma(const double *__restrict__ Yt, int K, int NP, int nSlices, double *G, const LmState *S = nullptr) {
    if (lm_skip(S, false)) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NT = NP / 16;
    // 1-D grid of tiles x slice groups.  Workgroups go round-robin over the 8 XCDs, each with its own L2: the mapping below gives every XCD
    // its own slice groups (all tiles of them), so that a row of Yt is fetched from memory by ONE L2 instead of by all eight
    const int nTiles = NT * (NT + 1) / 2, nGroups = nSlices / 4;
    int lin = blockIdx.x;
    if (nGroups % 8 == 0) { const int xcd = lin & 7, within = lin >> 3, gpx = nGroups / 8; lin = (xcd * gpx + within / nTiles) * nTiles + within % nTiles; }
    int t = lin % nTiles, tr = 0;                       // upper-triangle tile index -> (tr, tc)
    while (t >= NT - tr) { t -= NT - tr; tr++; }
    const int tc = tr + t;
    const int slice = (lin / nTiles) * 4 + wave;
    const int per = (((K + nSlices - 1) / nSlices) + 3) & ~3;
    const int k0 = slice * per, k1 = min(K, k0 + per);
    if (k0 >= k1) return;
    const int i = lane & 15, kk = lane >> 4;
    constexpr int C = 16;                               // 4-row steps per chunk
    const double *pa = Yt + tr * 16 + i, *pb = Yt + tc * 16 + i;
    double a0[C], b0[C], a1[C], b1[C];
    auto load = [&](double (&a)[C], double (&b)[C], int kb) {
#pragma unroll
        for (int u = 0; u < C; u++) {
            const size_t row = (size_t)min(kb + 4 * u + kk, k1 - 1);          // clamped: no branch around the loads
            a[u] = pa[row * NP]; b[u] = pb[row * NP];
        }
    };
    v4f64 acc = {0, 0, 0, 0};
    auto mma = [&](const double (&a)[C], const double (&b)[C], int kb) {
#pragma unroll
        for (int u = 0; u < C; u++) {
            const bool ok = kb + 4 * u + kk < k1;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ok ? a[u] : 0.0, ok ? b[u] : 0.0, acc, 0, 0, 0);
        }
    };
    load(a0, b0, k0);
    if (k0 + 4 * C < k1) load(a1, b1, k0 + 4 * C);
    for (int kb = k0; kb < k1; kb += 8 * C) {
        mma(a0, b0, kb);
        if (kb + 8 * C < k1) load(a0, b0, kb + 8 * C);
        if (kb + 4 * C < k1) {
            mma(a1, b1, kb + 4 * C);
            if (kb + 12 * C < k1) load(a1, b1, kb + 12 * C);
        }
    }
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
        const double g = acc[reg];
        if (g != 0.0) atomicAdd(&G[(size_t)(tr * 16 + kk + 4 * reg) * NP + tc * 16 + i], g);
    }
}

// 1/sqrt(d) from the hardware estimate plus two Newton steps (full double accuracy for the well-scaled pivots here); the
// IEEE sqrt and divide sequences are ~10x longer and sit on the serial pivot chain of the factorisation.
__device__ __forceinline__ double fast_rsqrt(double d) {
    double y = __builtin_amdgcn_rsq(d);
    y = y * (1.5 - 0.5 * d * y * y);
    y = y * (1.5 - 0.5 * d * y * y);
    return y;
}

// Reduced system (H_pp + lambda I - Y Y^T) x_p = b_p - Y z: blocked right-looking Cholesky (panel width 8) of the matrix
// augmented with the right-hand side as an extra ROW (its factor row is the forward substitution).  Per panel: wave 0
// factors the 8x8 diagonal block in registers and publishes it through LDS, one thread per row solves its 8 panel entries,
// then the trailing matrix takes the rank-8 update: three barriers per 8 columns.  The backward substitution keeps y in the
// registers of wave 0 and walks rows of L (contiguous), so its serial chain is a broadcast and one FMA per unknown.
// A lives in LDS when it fits (n <= 136), else in `Aglob` (L2).
constexpr int kPW = 8;
// USE_LDS is a template parameter so that every access keeps a static address space (a runtime select would make A a
// generic pointer and turn each LDS access into a flat_load / flat_store).
__device__ __forceinline__ double readlane_f64(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// The LM control flow lives on the host (it must poll the reference's stop flag between trials), and every trial ends with a decision on eight
// scalars.  Instead of a device-to-host copy plus hipStreamSynchronize (~30 us of runtime latency per trial) the scalars are PUBLISHED into
// fine-grained pinned host memory by a one-wave kernel, followed by a sequence number; the host spins on that number (a few us).
__global__ void k_ba_publish(const double *__restrict__ scal, volatile double *hostScal, unsigned long long seq) {
    if (threadIdx.x < 8) hostScal[threadIdx.x] = scal[threadIdx.x];
    __threadfence_system();
    __builtin_amdgcn_s_barrier();
    if (threadIdx.x == 0) __hip_atomic_store(reinterpret_cast<unsigned long long *>(const_cast<double *>(hostScal + 8)), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// First slot only, after chi2(estimate), the first buildSystem and k_ba_maxdiag: currentChi, and computeLambdaInit = tau * max |H_jj|
__global__ void k_lm_begin(const double *__restrict__ scal, LmState *S) {
    if (threadIdx.x != 0 || S->done) return;
    S->currentChi = scal[0]; S->iniChi = scal[0];
    S->lambda = 1e-5 * scal[2]; S->ni = 2; S->nBad = 0;
}
// The end of one LM trial (levenberg.cpp:102-166): rho from the trial's chi2 (scal[0]), the gain denominator (scal[1]) and the solver's verdict
// (scal[3]); accept -> the trial buffer becomes the estimate and the next slot rebuilds the system; reject -> larger damping, same system.
// Iteration bookkeeping as the host loop had it: at most 10 trials per iteration, stop after `maxIt` iterations, on rho == 0, or after three
// iterations in a row that gain less than a thousandth; the stop flag is looked at where the host loop looked at it.
__global__ void k_lm_decide(const double *__restrict__ scal, LmState *S, LmMirror *mirror, const volatile int32_t *stopWord, unsigned long long seq) {
    if (threadIdx.x != 0) return;
    if (!S->done) {
        const bool ok2 = S->nUnknowns == 0 || scal[3] != 0.0;
        const double tempChi = ok2 ? scal[0] : DBL_MAX;
        double rho = S->currentChi - tempChi;
        rho /= scal[1] + 1e-3;
        if (rho > 0 && isfinite(tempChi)) {
            const double tr = 2 * rho - 1;
            double alpha = 1. - tr * tr * tr;
            alpha = fmin(alpha, 2. / 3.);
            S->lambda *= fmax(1. / 3., alpha);
            S->ni = 2;
            S->currentChi = tempChi;
            S->cur ^= 1;                                   // discardTop(): the trial state becomes the estimate
        } else {
            S->lambda *= S->ni;
            S->ni *= 2;                                    // pop(): keep the current state
        }
        S->qmax++; S->trials++;
        const bool stop = __hip_atomic_load(const_cast<const int32_t *>(stopWord), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
        if (rho < 0 && S->qmax < 10 && !stop) {
            S->needBuild = 0;                              // another trial on the same system
        } else {
            S->iters++;
            bool term = S->qmax == 10 || rho == 0;
            if (!term) {
                if ((S->iniChi - S->currentChi) * 1e3 < S->iniChi) S->nBad++; else S->nBad = 0;
                term = S->nBad >= 3;
            }
            S->it++;
            if (term || S->it >= S->maxIt || stop) S->done = 1;
            else { S->needBuild = 1; S->qmax = 0; S->iniChi = S->currentChi; }
        }
    }
    mirror->cur = S->cur; mirror->iters = S->iters; mirror->trials = S->trials; mirror->done = S->done; mirror->trialsDone = S->trials;
    __threadfence_system();
    __hip_atomic_store(const_cast<unsigned long long *>(&mirror->seq), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <bool USE_LDS>
__global__ __launch_bounds__(1024) void k_ba_solve(BADev B, double lambda, const double *G, int NP, double *Aglob) {
    extern __shared__ double sa[];
    __shared__ double ldb[kPW * kPW], rdb[kPW];
    __shared__ int sFail;
    const int n = B.n, tid = threadIdx.x, nt = blockDim.x;
    const int ld = n + 1;
    auto A = [&]() { if constexpr (USE_LDS) return sa; else return Aglob; }();
    auto rdg = A + (size_t)(n + 1) * ld;                     // reciprocals of the factor's diagonal
    auto g = [&](int r, int c) -> double { return (r / 16 <= c / 16) ? G[(size_t)r * NP + c] : G[(size_t)c * NP + r]; };
    if (tid == 0) sFail = 0;
    for (int idx = tid; idx < (n + 1) * n; idx += nt) {
        const int i = idx / n, j = idx - i * n;
        if (j > i) continue;
        double v;
        if (i == n) v = B.bp[j] - g(j, n);
        else {
            v = -g(i, j);
            if (i / 6 == j / 6) v += B.Hpp[(size_t)(i / 6) * 36 + (i % 6) * 6 + (j % 6)];
            if (i == j) v += lambda;
        }
        A[(size_t)i * ld + j] = v;
    }
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += kPW) {
        const int w = min(kPW, n - c0);
        if (tid < 64) {                                      // wave 0: 8x8 diagonal block in registers (all lanes alike)
            double Ld[kPW][kPW], rd[kPW];
#pragma unroll
            for (int a = 0; a < kPW; a++)
#pragma unroll
                for (int b = 0; b < kPW; b++) Ld[a][b] = (a < w && b <= a) ? A[(size_t)(c0 + a) * ld + c0 + b] : (a == b ? 1.0 : 0.0);
            bool bad = false;
#pragma unroll
            for (int j = 0; j < kPW; j++) {
                double d = Ld[j][j];
#pragma unroll
                for (int k = 0; k < kPW; k++) if (k < j) d -= Ld[j][k] * Ld[j][k];
                if (!(d > 0) || !isfinite(d)) bad = true;
                const double rs = fast_rsqrt(d);
                rd[j] = rs;
                Ld[j][j] = d * rs;
#pragma unroll
                for (int i = 0; i < kPW; i++) if (i > j) {
                    double t = Ld[i][j];
#pragma unroll
                    for (int k = 0; k < kPW; k++) if (k < j) t -= Ld[i][k] * Ld[j][k];
                    Ld[i][j] = t * rs;
                }
            }
            // publish: lane (a*8+b) stores one entry of the block (static register selection)
            double mine = 0, myrd = 0;
#pragma unroll
            for (int a = 0; a < kPW; a++)
#pragma unroll
                for (int b = 0; b < kPW; b++) if (tid == a * kPW + b) mine = Ld[a][b];
#pragma unroll
            for (int a = 0; a < kPW; a++) if (tid == a) myrd = rd[a];
            ldb[tid] = mine;
            if (tid < kPW) { rdb[tid] = myrd; if (tid < w) rdg[c0 + tid] = myrd; }
            const int a = tid / kPW, b = tid - a * kPW;
            if (a < w && b <= a) A[(size_t)(c0 + a) * ld + c0 + b] = mine;    // nobody else reads the diagonal block now
            if (tid == 0 && bad) sFail = 1;
        }
        __syncthreads();
        if (sFail) break;
        // rows below the block: L[r][c0..] = A[r][c0..] * Ld^-T  (one thread per row)
        if (c0 + w + tid <= n) {                              // the block's factor, once per thread, into registers
            double lb[kPW][kPW], rb[kPW];
#pragma unroll
            for (int b = 0; b < kPW; b++) {
                rb[b] = rdb[b];
#pragma unroll
                for (int k = 0; k < kPW; k++) lb[b][k] = k < b ? ldb[b * kPW + k] : 0.0;
            }
        for (int r = c0 + w + tid; r <= n; r += nt) {
            double x[kPW];
#pragma unroll
            for (int b = 0; b < kPW; b++) x[b] = b < w ? A[(size_t)r * ld + c0 + b] : 0.0;
#pragma unroll
            for (int b = 0; b < kPW; b++) {
                double t = x[b];
#pragma unroll
                for (int k = 0; k < kPW; k++) if (k < b) t -= x[k] * lb[b][k];
                x[b] = t * rb[b];
            }
#pragma unroll
            for (int b = 0; b < kPW; b++) if (b < w) A[(size_t)r * ld + c0 + b] = x[b];
        }
        }
        __syncthreads();
        // trailing update: A[i][k] -= sum_q L[i][c0+q] L[k][c0+q] for c0+w <= k <= i <= n, k < n
        // 32 x 32 thread grid over the lower triangle: thread (ty, tx) owns rows i = t0 + ty + 32 a, columns k = t0 + tx + 32 b, k <= i;
        // the panel entries of its rows / columns are read once per (a, b) tile row / column, no integer division
        const int t0 = c0 + w, m = n + 1 - t0;
        const int ty = tid >> 5, tx = tid & 31;
        for (int i = t0 + ty; i <= n; i += 32) {
            double li[kPW];
#pragma unroll
            for (int q = 0; q < kPW; q++) li[q] = q < w ? A[(size_t)i * ld + c0 + q] : 0.0;
            for (int k = t0 + tx; k <= i && k < n; k += 32) {
                double acc = 0;
#pragma unroll
                for (int q = 0; q < kPW; q++) if (q < w) acc += li[q] * A[(size_t)k * ld + c0 + q];
                A[(size_t)i * ld + k] -= acc;
            }
        }
        (void)m;
        __syncthreads();
    }
    if (sFail) {
        if (tid == 0) B.scal[3] = 0.0;
        for (int i = tid; i < n; i += nt) B.x[i] = 0;
        return;
    }
    // backward substitution L^T x = y on wave 0: lane owns unknowns lane, lane+64, lane+128, lane+192
    if (tid < 64) {
        double y[4], x[4] = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; q++) { const int i = tid + 64 * q; y[q] = i < n ? A[(size_t)n * ld + i] : 0.0; }
        for (int j = n - 1; j >= 0; j--) {
            const int jq = j >> 6, jl = j & 63;
            double row[4];                                   // row j of L: independent of the chain, issued first
#pragma unroll
            for (int q = 0; q < 4; q++) { const int i = tid + 64 * q; row[q] = i < j ? A[(size_t)j * ld + i] : 0.0; }
            const double rdj = rdg[j];
            const double ysel = jq == 0 ? y[0] : jq == 1 ? y[1] : jq == 2 ? y[2] : y[3];
            const double xj = readlane_f64(ysel, jl) * rdj;        // jl is wave-uniform: a readlane, not an LDS-routed shuffle
#pragma unroll
            for (int q = 0; q < 4; q++) {
                y[q] -= row[q] * xj;
                if (q == jq && tid == jl) x[q] = xj;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) { const int i = tid + 64 * q; if (i < n) B.x[i] = x[q]; }
        if (tid == 0) B.scal[3] = 1.0;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The same system, tile formulation (NT = NP / 16 <= 11 tile rows, i.e. up to 29 optimised key-frames; one workgroup of 8 waves, everything
// in LDS).  The lower triangle is held as 16 x 16 tiles, tile (I, J) at (I (I + 1) / 2 + J) * 256, COLUMN-major inside a tile: with that
// layout every operand of v_mfma_f64_16x16x4_f64 below is 64 consecutive doubles per instruction (lane l <-> offset 64 s + l), so the rank-16
// trailing update of a tile costs 12 conflict-free ds_read_b64, 4 MFMAs and 4 ds_write_b64 where the panel-8 kernel above spent 8 LDS reads
// per ENTRY (it was bound by LDS bandwidth: 3.9 us of the 5.7 us per panel).  On gfx950 an f64 MFMA has the rate of the f64 vector FMA (64
// cycles per 16x16x4): the matrix cores are used here for their operand bandwidth, not for flops.
// Per panel p (16 columns):
//   * every wave that owns rows of the panel keeps the 16 diagonal rows in lanes 0..15 and 48 rows below the block in lanes 16..63, one row
//     per lane, 16 panel entries in registers, and runs the unblocked right-looking factorisation on them (panel_factor): the rows below are
//     solved by the very instructions that factor the block - no separate triangular solve - and the waves (on different SIMDs) do not wait
//     for each other.  The last wave feeds the rows of the identity through the same instructions and so obtains W_p = L_pp^-1, stored
//     transposed in place of the diagonal tile (nobody reads L_pp again);
//   * barrier; trailing tiles (I, J), p < J <= I, C^T -= L_J L_I^T on the matrix cores (transposed product: its result layout is again
//     lane l <-> offset 64 r + l); barrier.
// The right-hand side rides along as row n (its panel entries are y = L^-1 b).  Backward substitution by tiles with the W_p on one wave,
// no barrier (see there).  Measured on the 20 key-frame window (120 unknowns): 36 us per call against 104 us for the panel-8 kernel;
// 81 k cycles = assembly 10 k (two global round trips), eight panels x (load 0.5 k + factor 4.5 k, a dependent chain of ~270 cycles per
// pivot) 43 k, trailing updates 18 k, substitution 9 k.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kSolveTilesMax = 11;
constexpr int kSolveThreads = 512, kSolveWaves = kSolveThreads / 64;      // 8 waves: 256 registers each (the panel step and the substitution want ~150)
__device__ __forceinline__ int tile_at(int I, int J) { return (I * (I + 1) / 2 + J) * 256; }

// one 16-column panel on the rows a wave holds (one row per lane, `a` = its 16 panel entries): unblocked right-looking Cholesky, pivots and
// multipliers by v_readlane from lanes 0..15.  Written software-pipelined - column j + 1 is updated first and its pivot's reciprocal square root
// started before the rest of column j's rank-1 update - so that the update's instructions fill the latency of the pivot chain.
// sum over the four rows of 16 lanes (lanes l, l ^ 16, l ^ 32, l ^ 48), result in all of them: the gfx950 row-swap instructions, no LDS
__device__ __forceinline__ double rows_allreduce(double v) {
    unsigned lo = __double2loint(v), hi = __double2hiint(v);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    lo = __double2loint(v); hi = __double2hiint(v);
    a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}

// sum over the 16 lanes of a row, result in all of them: rotate-and-add by DPP
__device__ __forceinline__ double row_allreduce(double v) {
#define RUMI_ROR_ADD(ctl) v += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), ctl, 0xf, 0xf, false), \
                                                __builtin_amdgcn_update_dpp(0, __double2loint(v), ctl, 0xf, 0xf, false))
    RUMI_ROR_ADD(0x128); RUMI_ROR_ADD(0x124); RUMI_ROR_ADD(0x122); RUMI_ROR_ADD(0x121);     // row_ror:8, 4, 2, 1
#undef RUMI_ROR_ADD
    return v;
}

// 1 / sqrt(d) for the pivots of k_ba_solve_tiles: v_rsq_f64 (5e-8 relative, tools/rsq_probe.hip) and ONE Newton step (4e-15): the second step
// of fast_rsqrt buys 2.7e-16 for four more operations on the dependent chain of every pivot
__device__ __forceinline__ double fast_rsqrt1(double d) {
    const double y = __builtin_amdgcn_rsq(d);
    return y * __builtin_fma(-0.5 * d * y, y, 1.5);
}

// One 16-column panel on the rows a wave holds (one row per lane, `a` = its 16 panel entries; lanes 0..15 hold the diagonal block's rows):
// unblocked right-looking Cholesky.  The dependent chain of a column - scale, update the next two columns, next pivot, reciprocal square
// root - takes its pivot and multipliers from lanes 0..15 by v_readlane; the multipliers of the columns further right (not needed for two
// more steps) go through a 16-double LDS buffer of the wave as broadcast reads, issued one step ahead of their use: 2 instructions per
// pair less than readlane + hazard nop + fma, which is what bounds this single-wave loop.
// Columns j >= w (padding of the last panel) run through the same instructions with the pivot forced to 1: they hold finite values that only
// meet each other, so there is one straight-line instruction stream, no branch per column.
__device__ __forceinline__ bool panel_factor(double (&a)[16], int w, int lane, double *buf) {
    bool bad = false;
    double d = readlane_f64(a[0], 0);                          // w >= 1
    if (!(d > 0) || !isfinite(d)) bad = true;
    double rs = fast_rsqrt1(d);
    double mPrev[16], aPrev = 0.0;                             // multipliers and scaled column of the previous step (background work in flight)
#pragma unroll
    for (int j = 0; j < 16; j++) {
        double m[16];
        a[j] *= rs;                                            // lane j: d * rs = sqrt(d); lanes above the diagonal carry values nobody reads
        if (j + 3 < 16) {
            if (lane < 16) buf[lane] = a[j];
            // lanes talk through LDS inside one wave: the hardware keeps a wave's LDS operations in order, the fence tells the compiler that
            // the loads below see another lane's store (without it they are "unchanged memory" for the lanes that did not store)
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = j + 3; k < 16; k++) m[k] = buf[k];
        }
        if (j + 1 < 16) {
            a[j + 1] = __builtin_fma(-a[j], readlane_f64(a[j], j + 1), a[j + 1]);
            const double dn = readlane_f64(a[j + 1], j + 1);
            d = j + 1 < w ? dn : 1.0;
            if (!(d > 0) || !isfinite(d)) bad = true;
            rs = fast_rsqrt1(d);
        }
        if (j + 2 < 16) a[j + 2] = __builtin_fma(-a[j], readlane_f64(a[j], j + 2), a[j + 2]);
        if (j >= 1) {                                          // background of step j - 1: columns j + 2 .. 15
#pragma unroll
            for (int k = j + 2; k < 16; k++) a[k] = __builtin_fma(-aPrev, mPrev[k], a[k]);
        }
        aPrev = a[j];
#pragma unroll
        for (int k = j + 3; k < 16; k++) mPrev[k] = m[k];
    }
    return bad;
}

// What the factorisation reads and writes.  WIN = false: G is the row-major Gram matrix the atomics of k_ba_syrk_mfma filled (left zeroed for the
// next trial); WIN = true (ba_windows.inc): G holds the lower-triangle tiles in the tile layout itself, summed in a fixed order by k_baw_reduce.
struct SolveIO { int n, NP; const double *Hpp, *bp; double *x, *okFlag, *scalZero; };
template <bool WIN>
__device__ __forceinline__ void solve_tiles_core(const SolveIO &B, double lambda, double *__restrict__ G) {
    extern __shared__ double T[];                             // tiles | y[NT*16]
    __shared__ int sFail;
    __shared__ double sBuf[kSolveWaves][16];                           // per wave: the scaled pivot column of the panel step in flight
    const int n = B.n, NP = B.NP, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NT = NP / 16;                                   // tile rows (the right-hand side is row n)
    const int PT = (n + 15) / 16;                             // tile columns that hold pivots
    const int nT = NT * (NT + 1) / 2;
    double *ys = T + (size_t)nT * 256;
    if (tid == 0) { sFail = 0; if (!WIN) { B.scalZero[0] = 0.0; B.scalZero[1] = 0.0; } }      // chi2 and scale of the trial are accumulated by the kernels after this one
    // ---- assemble: element (i, j) = -G[j][i] (+ b_p in row n), one wave per tile and all of a wave's loads in flight at once (G was just
    // written: one L2 latency, not one per tile); then the 6x6 blocks of H_pp + lambda I are added by one thread per entry
    {
        constexpr int kU = (kSolveTilesMax * (kSolveTilesMax + 1) / 2 + kSolveWaves - 1) / kSolveWaves;
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const int m = lane & 15, cq = lane >> 4;
        double v[kU][4];
        double hpp[(kSolveTilesMax * 16 * 6 + kSolveThreads - 1) / kSolveThreads];     // the 6x6 blocks: n * 6 entries, loaded early
#pragma unroll
        for (int u = 0; u < (int)(sizeof(hpp) / sizeof(double)); u++) hpp[u] = B.Hpp[min(tid + kSolveThreads * u, n * 6 - 1)];
#pragma unroll
        for (int u = 0; u < kU; u++) {
            const int t = wv + kSolveWaves * u;
            if (t < nT) {
                int I = 0, r = t;
                while (r > I) { r -= I + 1; I++; }
                const int i = I * 16 + m;
#pragma unroll
                for (int q = 0; q < 4; q++) {                  // loads without a branch around them (clamped addresses): all in flight together
                    const int j = r * 16 + cq + 4 * q, jc = min(j, n - 1);
                    double gv;
                    if constexpr (WIN) gv = G[(size_t)t * 256 + (cq + 4 * q) * 16 + m];       // tile (I, r), element (row i, column j) at (j & 15) * 16 + (i & 15)
                    else gv = G[(size_t)jc * NP + min(i, n)];
                    const double bv = B.bp[jc];
                    v[u][q] = (i <= n && j < n) ? (i == n ? bv - gv : -gv) : 0.0;
                    if constexpr (!WIN) { if (i <= n && j < n) G[(size_t)j * NP + i] = 0.0; }      // G is an accumulator of atomics: left zeroed for the next trial
                }
            }
        }
#pragma unroll
        for (int u = 0; u < kU; u++) {
            const int t = wv + kSolveWaves * u;
            if (t < nT) {
#pragma unroll
                for (int q = 0; q < 4; q++) T[t * 256 + (cq + 4 * q) * 16 + m] = v[u][q];
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < (int)(sizeof(hpp) / sizeof(double)); u++) {
            const int e = tid + kSolveThreads * u;
            if (e >= n * 6) break;
            const int kf = e / 36, ab = e - kf * 36, a6 = ab / 6, b6 = ab - a6 * 6;
            const int i = kf * 6 + a6, j = kf * 6 + b6;
            if ((i >> 4) >= (j >> 4)) {                        // the entry lies in a stored tile (upper entries of diagonal tiles included)
                const double h = hpp[u] + (a6 == b6 ? lambda : 0.0);
                T[tile_at(i >> 4, j >> 4) + (j & 15) * 16 + (i & 15)] += h;
            }
        }
    }
    __syncthreads();
    // One tile of the rank-16 trailing update of panel p: C(I, J)^T -= L(J, p) L(I, p)^T on the matrix cores
    auto update_tile = [&](int I, int J, int p) {
        const double *LJ = T + tile_at(J, p), *LI = T + tile_at(I, p);
        double *C = T + tile_at(I, J);
        v4f64 acc;
#pragma unroll
        for (int q = 0; q < 4; q++) acc[q] = C[64 * q + lane];
#pragma unroll
        for (int s = 0; s < 4; s++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-LJ[64 * s + lane], LI[64 * s + lane], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; q++) C[64 * q + lane] = acc[q];
    };
    // Panels with look-ahead.  Per panel p: [rows of the panel into registers] barrier [the owning waves factor it | the OTHER waves finish the
    // trailing update of panel p - 1 on the columns right of p] barrier [all waves: panel p applied to tile column p + 1 only] barrier.  The
    // factorisation (a dependent chain of ~270 cycles per pivot on a few waves) thus runs beside the bulk of the previous panel's update instead
    // of after it; only one tile column per panel is on the critical path.
    // Wave u < kSolveWaves - 1 owns the diagonal rows (lanes 0..15) and rows p*16 + 16 + 48 u + (lane - 16) below the block.  The last wave runs
    // the same instructions on the diagonal rows and, in lanes 16..31, on the rows of the identity: "solving" e_c^T against L_pp^T leaves
    // column c of L_pp^-1 in lane 16 + c - the inverse the backward substitution wants, for free.
    for (int p = 0; p < PT; p++) {
        const int w = min(16, n - p * 16);
        const int rowsBelow = NT * 16 - (p + 1) * 16;
        const bool inv = wave == kSolveWaves - 1;
        const bool mine = inv || wave == 0 || wave * 48 < rowsBelow;
        const int rrel = lane < 16 ? lane : 16 + 48 * wave + (lane - 16);         // row relative to the panel start
        const int Ip = p + (rrel >> 4);
        const bool live = mine && (lane < 16 || (!inv && Ip < NT));
        double *src = T + tile_at(live ? Ip : p, p) + (rrel & 15);
        double a[16];
        if (mine) {
#pragma unroll
            for (int c = 0; c < 16; c++) a[c] = src[c * 16];   // unconditional (the address is always inside the tile array): 16 reads in flight
#pragma unroll
            for (int c = 0; c < 16; c++) a[c] = live ? a[c] : (inv && lane == 16 + c ? 1.0 : 0.0);
        }
        __syncthreads();                                      // every wave holds the diagonal rows before the tile is overwritten
        if (mine) {
            const bool bad = panel_factor(a, w, lane, sBuf[wave]);
            if (inv) {
                // W^T in place of the diagonal tile: element (i, c) of W = L_pp^-1 at i * 16 + c; zero beyond w, so that the padding unknowns come out 0
                if (lane >= 16 && lane < 32) {
                    const int c = lane - 16;
#pragma unroll
                    for (int k = 0; k < 16; k++) T[tile_at(p, p) + k * 16 + c] = (k < w && c < w) ? a[k] : 0.0;
                }
            } else if (live && lane >= 16) {
#pragma unroll
                for (int c = 0; c < 16; c++) src[c * 16] = a[c];
            }
            if (!inv && live && p * 16 + rrel == n && (lane >= 16 || wave == 0)) {     // the right-hand side row: y = L^-1 b for these 16 columns
#pragma unroll
                for (int c = 0; c < 16; c++) ys[p * 16 + c] = c < w ? a[c] : 0.0;
            }
            if (wave == 0 && bad && lane == 0) sFail = 1;
        }
        int freeIdx = 0, nFree = 0;
        for (int u = 1; u < kSolveWaves - 1; u++) {
            const bool owns = u * 48 < rowsBelow;
            if (!owns) { if (u < wave) freeIdx++; nFree++; }
        }
        if (nFree == 0) { freeIdx = wave; nFree = kSolveWaves; }      // (cannot happen up to 11 tile rows: at most three waves own rows below the block)
        if (p > 0 && (!mine || nFree == kSolveWaves)) {
            // the rest of panel p - 1's update: tiles (I, J), p < J <= I < NT, shared among the waves that own no row of panel p
            const int m1 = NT - 1 - p, nTiles = m1 * (m1 + 1) / 2;
            for (int t = freeIdx; t < nTiles; t += nFree) {
                int Ir = 0, r = t;
                while (r > Ir) { r -= Ir + 1; Ir++; }
                update_tile(p + 1 + Ir, p + 1 + r, p - 1);
            }
        }
        __syncthreads();
        if (sFail) break;
        // panel p applied to tile column p + 1 (the next panel): tiles (I, p + 1), p < I < NT
        if (p + 1 < PT) {
            for (int I = p + 1 + wave; I < NT; I += kSolveWaves) update_tile(I, p + 1, p);
        }
        __syncthreads();
    }
    if (sFail) {
        if (tid == 0) *B.okFlag = 0.0;
        for (int i = tid; i < n; i += kSolveThreads) B.x[i] = 0;
        return;
    }
    // ---- backward substitution L^T x = y on wave 0, right-looking by tiles, no barrier: x_p = W_p^T (y_p - sum_{q > p} L(q, p)^T x_q).
    // Two lane layouts alternate so that no value has to be fetched from another lane by address:
    //   A  lane (c = lane & 15, g = lane >> 4) holds x_q[c]                                  (every g alike)
    //   B  lane (*, g) holds the four values [4 g .. 4 g + 3] of a 16-vector                (every lane of the row alike)
    // L(q, p)^T x_q:  lane (k = c, g) forms L[k][4 g + j] x_q[k] (A) and the sum over k is a rotate-and-add inside the row of 16 lanes -> B;
    // W_p^T r:        lane (c, g) forms sum_j W[4 g + j][c] r[4 g + j] (B) and the sum over g is two v_permlane*_swap exchanges  -> A.
    // As soon as x_q exists its products with ALL tiles (q, p < q) are accumulated (acc[p], unreduced), the tile (q, q - 1) first: only that
    // one, one rotate-and-add reduction and W are on the dependent chain of a step.
    if (wave == 0) {
        const int c = lane & 15, g = lane >> 4;
        double acc[kSolveTilesMax][4];
#pragma unroll
        for (int p = 0; p < kSolveTilesMax; p++) acc[p][0] = acc[p][1] = acc[p][2] = acc[p][3] = 0.0;
#pragma unroll
        for (int q = kSolveTilesMax - 1; q >= 0; q--) {
            if (q < PT) {
                // x_q from acc[q]
                const double *yq = ys + q * 16 + 4 * g, *Wt = T + tile_at(q, q) + (4 * g) * 16 + c;
                double xp = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) xp = __builtin_fma(Wt[j * 16], yq[j] - row_allreduce(acc[q][j]), xp);
                const double x = rows_allreduce(xp);
                if (g == 0 && q * 16 + c < n) B.x[q * 16 + c] = x;
                // its products with the tiles to the left, nearest first
#pragma unroll
                for (int p = kSolveTilesMax - 2; p >= 0; p--) {
                    if (p < q) {
                        const double *Lt = T + tile_at(q, p) + (4 * g) * 16 + c;
#pragma unroll
                        for (int j = 0; j < 4; j++) acc[p][j] = __builtin_fma(Lt[j * 16], x, acc[p][j]);
                    }
                }
            }
        }
        if (lane == 0) *B.okFlag = 1.0;
    }
}

__global__ __launch_bounds__(kSolveThreads) void k_ba_solve_tiles(BADev B, double lambda, double *__restrict__ G, int NP, const LmState *S = nullptr) {
    if (S) { if (S->done) return; lambda = S->lambda; }
    const SolveIO io{B.n, NP, B.Hpp, B.bp, B.x, B.scal + 3, B.scal};
    solve_tiles_core<false>(io, lambda, G);
}


// ==================================================================================================================
// Reduced system of a LARGE window (more than 42 optimised key-frames: global bundle adjustment after a loop closure or a map merge,
// Optimizer.cc:48-351).  The dense panel Y of the small-window path would be 3 P x 6 K; here the Schur complement is accumulated block-sparsely
// (a landmark seen by k key-frames touches k (k + 1) / 2 blocks of 6 x 6) into the dense lower triangle of the augmented matrix
// [H_pp + lambda I - sum W W^T ; (b_p - sum W z)^T], which a multi-workgroup blocked Cholesky (panel 64) then factors in place: per panel
// one wave factors the diagonal block in registers (row per lane, pivots and columns by v_readlane), one lane per row solves the rows
// below, and 64 x 64 tiles take the trailing update; the right-hand side rides along as the extra row, a blocked backward substitution
// finishes.  No host round trip inside a trial.
// ==================================================================================================================
constexpr int kNB = 64;

__global__ void k_big_init(BADev B, double lambda, double *A, int ld) {
    const int n = B.n;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)(n + 1) * n) return;
    const int i = (int)(idx / n), j = (int)(idx - (size_t)i * n);
    double v = 0;
    if (i == n) v = B.bp[j];
    else if (i / 6 == j / 6) { v = B.Hpp[(size_t)(i / 6) * 36 + (i % 6) * 6 + (j % 6)]; if (i == j) v += lambda; }
    A[(size_t)i * ld + j] = v;
    if (idx == 0) B.scal[3] = 1.0;
}

// W_e = H_pl(e) L_p (6 x 3), L_p L_p^T = (H_ll + lambda I)^-1
__global__ void k_big_w(BADev B, const double *Lp, double *W, int32_t *colOf) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B.nE) return;
    const int col = B.poseCol[B.eKF[e]];
    colOf[e] = col;
    if (col < 0) return;
    const double *L = Lp + (size_t)B.eMP[e] * 6, *h = B.Hpl + (size_t)e * 18;
    const double l00 = L[0], l10 = L[1], l20 = L[2], l11 = L[3], l21 = L[4], l22 = L[5];
    double *w = W + (size_t)e * 18;
#pragma unroll
    for (int a = 0; a < 6; a++) {
        const double h0 = h[a * 3], h1 = h[a * 3 + 1], h2 = h[a * 3 + 2];
        w[a * 3] = h0 * l00 + h1 * l10 + h2 * l20; w[a * 3 + 1] = h1 * l11 + h2 * l21; w[a * 3 + 2] = h2 * l22;
    }
}

// One wave per non-empty 6 x 6 block (ca, cb <= ca) of the Schur complement: the host groups the observation pairs (a, b) of all landmarks by
// block once per call (the structure is the same for every trial), lane (r, c) accumulates sum_pairs W_a[r] . W_b[c] in a register and
// subtracts it from the matrix with a plain store; lanes 36..41 of the diagonal blocks do the same for W_a z.  No atomics: device-scope f64
// atomics are served past the per-XCD L2s (the per-landmark atomic formulation measured 0.69 ms at 130 key-frames, LDS f64 atomics on a
// row strip per key-frame 0.9 - 1.6 ms, this one 0.05 ms).
constexpr int kSchurSeg = 32;   // pairs per wave: long blocks (the diagonal ones: every observation of the key-frame) are cut into segments
__global__ __launch_bounds__(256) void k_big_schur(BADev B, const int32_t *blk, int nb, const int32_t *pairs, const double *W, const double *z, double *A, int ld) {
    const int wv = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wv >= nb) return;
    const int ca = blk[4 * wv], cbm = blk[4 * wv + 1], s = blk[4 * wv + 2], len = blk[4 * wv + 3] - s;
    const int cb = cbm & 0x3fffffff;
    const bool multi = (cbm >> 30) != 0;                  // the block has more segments: combine with atomics
    // the segment's pairs, one per lane (coalesced), broadcast below: no dependent index loads inside the loop
    int myA = 0, myB = 0;
    if (lane < len) { myA = pairs[2 * (size_t)(s + lane)]; myB = pairs[2 * (size_t)(s + lane) + 1]; }
    const bool ent = lane < 36, rhs = lane >= 36 && lane < 42 && ca == cb;
    const int r = ent ? lane / 6 : rhs ? lane - 36 : 0, c = ent ? lane - (lane / 6) * 6 : 0;
    double acc = 0;
#pragma unroll 1
    for (int t = 0; t < len; t++) {
        const int ea = __builtin_amdgcn_readlane(myA, t), eb = __builtin_amdgcn_readlane(myB, t);
        const double *wa = W + (size_t)ea * 18 + r * 3, *wb = W + (size_t)eb * 18 + c * 3;
        acc += wa[0] * wb[0] + wa[1] * wb[1] + wa[2] * wb[2];
    }
    if (rhs) {                                            // diagonal block: pairs are (a, a); - W_a z on lanes 36..41
        acc = 0;
        for (int t = 0; t < len; t++) {
            const int ea = __builtin_amdgcn_readlane(myA, t);
            const double *wa = W + (size_t)ea * 18 + r * 3, *zp = z + 3 * (size_t)B.eMP[ea];
            acc += wa[0] * zp[0] + wa[1] * zp[1] + wa[2] * zp[2];
        }
    }
    if (!ent && !rhs) return;
    double *dst = ent ? &A[(size_t)(6 * ca + r) * ld + 6 * cb + c] : &A[(size_t)B.n * ld + 6 * ca + r];
    if (multi) atomicAdd(dst, -acc); else *dst -= acc;
}


// diagonal block [j0, j0 + w) in LDS, sub-panels of 8 columns: wave 0 factors the 8 x 8 sub-diagonal in registers (every lane the same
// values: a chain of 36 dependent steps instead of 8 LDS round trips), lane i solves row i against it and publishes the row's eight entries;
// then four waves (lane = row, 16 columns each) take the trailing update with 8-term dot products.  Rows past w carry an identity; the
// update also runs over the unused upper triangle, which keeps the loops uniform.
__global__ __launch_bounds__(256) void k_chol_diag(double *A, int ld, int n, int j0, double *rdg, double *scal) {
    constexpr int kS = kNB + 1, kP = 8;
    __shared__ double S[kNB * kS];
    __shared__ double P[kNB * kP];                         // the current sub-panel, row-major: P[i][q] = L[i][c0 + q]
    const int tid = threadIdx.x, lane = tid & 63, part = tid >> 6, w = min(kNB, n - j0);
    for (int idx = tid; idx < kNB * kNB; idx += 256) {
        const int r = idx / kNB, c = idx - r * kNB;
        S[r * kS + c] = (r < w && c <= r) ? A[(size_t)(j0 + r) * ld + j0 + c] : (r == c ? 1.0 : 0.0);
    }
    __syncthreads();
    bool bad = false;
    double myRs = 1.0;
    double *__restrict__ row = S + lane * kS;
    for (int c0 = 0; c0 < kNB; c0 += kP) {
        if (part == 0) {
            double Ld[kP][kP], rd[kP];
#pragma unroll
            for (int a = 0; a < kP; a++)
#pragma unroll
                for (int b = 0; b < kP; b++) Ld[a][b] = b <= a ? S[(c0 + a) * kS + c0 + b] : 0.0;
#pragma unroll
            for (int j = 0; j < kP; j++) {
                double d = Ld[j][j];
#pragma unroll
                for (int k = 0; k < j; k++) d -= Ld[j][k] * Ld[j][k];
                if (!(d > 0) || !isfinite(d)) bad = true;
                const double rs = fast_rsqrt(d);
                rd[j] = rs;
                Ld[j][j] = d * rs;
#pragma unroll
                for (int i = j + 1; i < kP; i++) {
                    double t = Ld[i][j];
#pragma unroll
                    for (int k = 0; k < j; k++) t -= Ld[i][k] * Ld[j][k];
                    Ld[i][j] = t * rs;
                }
            }
            // own row: rows of the sub-diagonal take their factor row, rows below solve x Ld^T = a, rows above keep zeros
            double x[kP];
#pragma unroll
            for (int b = 0; b < kP; b++) x[b] = row[c0 + b];
            const int a = lane - c0;
#pragma unroll
            for (int b = 0; b < kP; b++) {
                double t = x[b];
#pragma unroll
                for (int k = 0; k < b; k++) t -= x[k] * Ld[b][k];
                x[b] = t * rd[b];
            }
#pragma unroll
            for (int q = 0; q < kP; q++) {
#pragma unroll
                for (int b = 0; b < kP; b++) if (a == q) { x[b] = b <= q ? Ld[q][b] : 0.0; if (b == q) myRs = rd[q]; }
            }
            if (a < 0) {
#pragma unroll
                for (int b = 0; b < kP; b++) x[b] = 0.0;
            }
#pragma unroll
            for (int b = 0; b < kP; b++) { if (a >= 0) row[c0 + b] = x[b]; P[lane * kP + b] = x[b]; }
        }
        __syncthreads();
        if (part * 16 + 15 >= c0 + kP) {                  // this wave's 16 columns of the trailing block
            double li[kP];
#pragma unroll
            for (int q = 0; q < kP; q++) li[q] = P[lane * kP + q];
#pragma unroll 4
            for (int u = 0; u < 16; u++) {
                const int k = part * 16 + u;
                double acc = 0;
#pragma unroll
                for (int q = 0; q < kP; q++) acc += li[q] * P[k * kP + q];
                if (k >= c0 + kP) row[k] -= acc;
            }
        }
        __syncthreads();
    }
    for (int idx = tid; idx < kNB * kNB; idx += 256) {
        const int r = idx / kNB, c = idx - r * kNB;
        if (r < w && c <= r) A[(size_t)(j0 + r) * ld + j0 + c] = S[r * kS + c];
    }
    if (part == 0 && lane < w) rdg[j0 + lane] = myRs;
    if (bad && tid == 0) scal[3] = 0.0;
}

// rows below the block (the right-hand side row n included): L[r][j0..] = A[r][j0..] Ld^-T, one lane per row.  Eight columns at a time live
// in registers; the columns already solved are read back from LDS (column-major: conflict-free), the block's factor as LDS broadcasts
// (stored transposed, so the eight factors of one step are contiguous), the reciprocal pivots on its diagonal.
__global__ __launch_bounds__(256) void k_chol_trsm(double *A, int ld, int n, int j0, const double *rdg) {
    __shared__ double sLt[kNB * kNB], sX[kNB * 64];          // sLt[k][b] = L[b][k]
    const int tid = threadIdx.x, lane = tid & 63, w = min(kNB, n - j0), t0 = j0 + w;
    for (int k = tid >> 6; k < kNB; k += 4) {                 // lanes over b: conflict-free LDS rows (the 64 x 64 block is re-read from L1 / L2)
        const int b = lane;
        sLt[k * kNB + b] = (b < w && k < b) ? A[(size_t)(j0 + b) * ld + j0 + k] : (k == b) ? (b < w ? rdg[j0 + b] : 1.0) : 0.0;
    }
    const int r0 = t0 + blockIdx.x * 64;
    // the 64 rows of this workgroup, coalesced; element (row rr, column c) sits at sX[c][rr ^ c]: the fill (lanes over c) and the solve
    // (lanes over rr) are both conflict-free
    for (int idx = tid; idx < 64 * kNB; idx += 256) {
        const int rr = idx / kNB, c = idx - rr * kNB;
        sX[c * 64 + (rr ^ c)] = (r0 + rr <= n && c < w) ? A[(size_t)(r0 + rr) * ld + j0 + c] : 0.0;
    }
    __syncthreads();
    if (tid < 64)
    for (int sp = 0; sp < kNB / 8; sp++) {
        const int c0 = sp * 8;
        double x[8];
#pragma unroll
        for (int b = 0; b < 8; b++) x[b] = sX[(c0 + b) * 64 + (lane ^ (c0 + b))];
#pragma unroll 4
        for (int k = 0; k < c0; k++) {
            const double xk = sX[k * 64 + (lane ^ k)];
            const double *l = sLt + k * kNB + c0;
#pragma unroll
            for (int b = 0; b < 8; b++) x[b] -= xk * l[b];
        }
#pragma unroll
        for (int b = 0; b < 8; b++) {
            double t = x[b];
#pragma unroll
            for (int k = 0; k < b; k++) t -= x[k] * sLt[(c0 + k) * kNB + c0 + b];
            x[b] = t * sLt[(c0 + b) * kNB + c0 + b];
        }
#pragma unroll
        for (int b = 0; b < 8; b++) sX[(c0 + b) * 64 + (lane ^ (c0 + b))] = x[b];
    }
    __syncthreads();
    for (int idx = tid; idx < 64 * kNB; idx += 256) {
        const int rr = idx / kNB, c = idx - rr * kNB;
        if (r0 + rr <= n && c < w) A[(size_t)(r0 + rr) * ld + j0 + c] = sX[c * 64 + (rr ^ c)];
    }
}

// trailing update A[i][k] -= sum_q L[i][j0 + q] L[k][j0 + q] for t0 <= k <= i <= n, k < n: one 64 x 64 tile per workgroup, 4 x 4 per thread
__global__ __launch_bounds__(256) void k_chol_syrk(double *A, int ld, int n, int j0) {
    const int w = min(kNB, n - j0), t0 = j0 + w;
    const int ti = blockIdx.y, tk = blockIdx.x;
    if (tk > ti) return;
    const int i0 = t0 + ti * 64, k0 = t0 + tk * 64;
    if (k0 >= n) return;
    __shared__ double sI[64][33], sK[64][33];
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = 0;
    for (int q0 = 0; q0 < w; q0 += 32) {
        for (int idx = tid; idx < 64 * 32; idx += 256) {
            const int r = idx >> 5, q = idx & 31;
            const bool qv = q0 + q < w;
            sI[r][q] = (qv && i0 + r <= n) ? A[(size_t)(i0 + r) * ld + j0 + q0 + q] : 0.0;
            sK[r][q] = (qv && k0 + r < n) ? A[(size_t)(k0 + r) * ld + j0 + q0 + q] : 0.0;
        }
        __syncthreads();
#pragma unroll 8
        for (int q = 0; q < 32; q++) {
            double li[4], lk[4];
#pragma unroll
            for (int a = 0; a < 4; a++) { li[a] = sI[ty + 16 * a][q]; lk[a] = sK[tx + 16 * a][q]; }
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) acc[a][b] += li[a] * lk[b];
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int i = i0 + ty + 16 * a, k = k0 + tx + 16 * b;
            if (i <= n && k < n && k <= i) A[(size_t)i * ld + k] -= acc[a][b];
        }
}

// L^T x = y (y = the factor's row n), blocks of 64 from the back: wave 0 solves the block, all threads update the unknowns before it
__global__ __launch_bounds__(1024) void k_chol_backsub(const double *A, int ld, int n, const double *rdg, double *x, const double *scal) {
    extern __shared__ double ys[];
    __shared__ double sD[kNB * kNB];
    const int tid = threadIdx.x;
    if (scal[3] == 0.0) {                               // not positive definite: g2o's solve() fails, the LM step is rejected
        for (int i = tid; i < n; i += 1024) x[i] = 0;
        return;
    }
    for (int i = tid; i < n; i += 1024) ys[i] = A[(size_t)n * ld + i];
    __syncthreads();
    for (int jb = (n + kNB - 1) / kNB - 1; jb >= 0; jb--) {
        const int j0 = jb * kNB, w = min(kNB, n - j0);
        for (int idx = tid; idx < kNB * kNB; idx += 1024) {   // the diagonal block, coalesced, for the sequential solve below
            const int r = idx / kNB, c = idx - r * kNB;
            sD[idx] = (r < w && c < r) ? A[(size_t)(j0 + r) * ld + j0 + c] : 0.0;
        }
        __syncthreads();
        if (tid < 64) {                                       // lane t carries unknown j0 + t; the solved one is broadcast by readlane
            double y = tid < w ? ys[j0 + tid] : 0.0;
            const double rd = tid < w ? rdg[j0 + tid] : 0.0;
            for (int jj = w - 1; jj >= 0; jj--) {
                const double l = sD[jj * kNB + tid];          // row jj of the block (zero from the diagonal on)
                const double xj = readlane_f64(y, jj) * readlane_f64(rd, jj);
                y = tid == jj ? xj : y - l * xj;
            }
            if (tid < w) ys[j0 + tid] = y;
        }
        __syncthreads();
        for (int i = tid; i < j0; i += 1024) {
            double acc0 = 0, acc1 = 0;
            const double *col = A + (size_t)j0 * ld + i;
            int q = 0;
#pragma unroll 1
            for (; q + 16 <= w; q += 16) {
                double v[16];
#pragma unroll
                for (int u = 0; u < 16; u++) v[u] = col[(size_t)(q + u) * ld];
#pragma unroll
                for (int u = 0; u < 16; u += 2) { acc0 += v[u] * ys[j0 + q + u]; acc1 += v[u + 1] * ys[j0 + q + u + 1]; }
            }
            for (; q < w; q++) acc0 += col[(size_t)q * ld] * ys[j0 + q];
            ys[i] -= acc0 + acc1;
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += 1024) x[i] = ys[i];
}

// x_l = D^-1 (b_l - H_pl^T x_p); trial state = oplus(current, x); scale += x^T (lambda x + b)
// Landmark back-substitution x_l = D^-1 (b_l - H_pl^T x_p) (block_solver.hpp:468-481), oplus of points and poses, and the
// gain-ratio denominator.  Eight lanes share a landmark (its ~15 edges are two rounds instead of fifteen dependent ones); the
// first nKF * 8 lanes past the landmarks carry the poses (one per group of eight).
constexpr int kLmLanes = 8;
__global__ __launch_bounds__(256) void k_ba_update(BADev B, double lambda, const double *T, const double *X, double *Tt, double *Xt, const LmState *S = nullptr) {
    __shared__ double red[4];
    if (S) { if (S->done) return; lambda = S->lambda; T = S->T[S->cur]; X = S->X[S->cur]; Tt = S->T[S->cur ^ 1]; Xt = S->X[S->cur ^ 1]; }
    const int gi = (blockIdx.x * 256 + threadIdx.x) / kLmLanes, sub = threadIdx.x & (kLmLanes - 1);
    double acc[1] = {0};
    if (gi < B.nMP) {
        const int p = gi;
        double c0 = 0, c1 = 0, c2 = 0;
        for (int s = B.ptStart[p] + sub; s < B.ptStart[p + 1]; s += kLmLanes) {
            const int e = B.ptEdge[s], col = B.poseCol[B.eKF[e]];
            if (col < 0) continue;
            const double *h = B.Hpl + (size_t)e * 18, *xp = B.x + col * 6;
#pragma unroll
            for (int a = 0; a < 6; a++) { c0 -= h[a * 3] * xp[a]; c1 -= h[a * 3 + 1] * xp[a]; c2 -= h[a * 3 + 2] * xp[a]; }
        }
#pragma unroll
        for (int o = kLmLanes / 2; o > 0; o >>= 1) { c0 += __shfl_xor(c0, o, kLmLanes); c1 += __shfl_xor(c1, o, kLmLanes); c2 += __shfl_xor(c2, o, kLmLanes); }
        if (sub == 0) {
            const double b0 = B.bl[3 * p], b1 = B.bl[3 * p + 1], b2 = B.bl[3 * p + 2];
            c0 += b0; c1 += b1; c2 += b2;
            const double *I = B.Dinv + (size_t)p * 9;
            const double x0 = I[0] * c0 + I[1] * c1 + I[2] * c2, x1 = I[3] * c0 + I[4] * c1 + I[5] * c2, x2 = I[6] * c0 + I[7] * c1 + I[8] * c2;
            B.x[B.n + 3 * p] = x0; B.x[B.n + 3 * p + 1] = x1; B.x[B.n + 3 * p + 2] = x2;
            Xt[3 * p] = X[3 * p] + x0; Xt[3 * p + 1] = X[3 * p + 1] + x1; Xt[3 * p + 2] = X[3 * p + 2] + x2;
            acc[0] = x0 * (lambda * x0 + b0) + x1 * (lambda * x1 + b1) + x2 * (lambda * x2 + b2);
        }
    } else if (gi < B.nMP + B.nKF && sub == 0) {
        const int k = gi - B.nMP, col = B.poseCol[k];
        DSE3 P = load_pose(T, k);
        if (col >= 0) {
            const double *xp = B.x + col * 6;
            double u[6];
            for (int a = 0; a < 6; a++) { u[a] = xp[a]; acc[0] += xp[a] * (lambda * xp[a] + B.bp[col * 6 + a]); }
            P = se3_mul(se3_exp(u), P);
        }
        store_pose(Tt, k, P);
    }
    block_sum<1>(acc, red);
    if (threadIdx.x == 0 && acc[0] != 0.0) atomicAdd(&B.scal[1], acc[0]);
}

// merge BA, between its two optimisations (Optimizer.cc:3996-4010): edges with chi2 > 5.991 or non-positive depth go to level 1
__global__ void k_ba_mark(BADev B, const double *T, const double *X, uint8_t *off) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B.nE) return;
    const int p = B.eMP[e];
    const D3 pc = se3_map(load_pose(T, B.eKF[e]), D3{X[3 * p], X[3 * p + 1], X[3 * p + 2]});
    off[e] = (B.lastChi2[e] > 5.991 || !(pc.z > 0.0)) ? 1 : 0;
}

// erase flags of the edges (Optimizer.cc:1292) and, in the same launch, the final state gathered behind them: [T | X | erase] leaves in one copy
__global__ void k_ba_finalize(BADev B, const double *T, const double *X, int useLast, uint8_t *erase, double *outT, double *outX) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < B.nKF * 8) outT[e] = T[e];
    if (e < B.nMP * 3) outX[e] = X[e];
    if (e >= B.nE) return;
    const int p = B.eMP[e];
    const D3 pc = se3_map(load_pose(T, B.eKF[e]), D3{X[3 * p], X[3 * p + 1], X[3 * p + 2]});
    double chi2 = B.lastChi2[e];
    if (!useLast) {
        double u, v;
        cam_project(B.cam, pc, u, v);
        const double e0 = (double)B.obs[2 * e] - u, e1 = (double)B.obs[2 * e + 1] - v, w = (double)B.info[e];
        chi2 = e0 * w * e0 + e1 * w * e1;
    }
    erase[e] = (chi2 > 5.991 || !(pc.z > 0.0)) ? 1 : 0;       // Optimizer.cc:1292
}

// Sim3Solver::ComputeInliersNum (R/lib_src/Sim3Solver.cc:564-664): one lane per matched key-point pair.
// g2o::Sim3::map = s * (r * xyz) + t in double (G/types/sim3.h:144-146), Pinhole::project(Vector3d) in double then .cast<float>()
// (Pinhole.cpp:35-41), squared reprojection errors in float, tests against 2 * 9.210 * mvLevelSigma2 in double.
__global__ void k_sim3_inliers(int total, const int32_t *pairOf, const double *Sc1w2, const double *Sc2w1, const float *K1, const float *K2,
                               const float *X1, const float *X2, const float *kp1, const float *kp2, const float *sigma1, const float *sigma2,
                               const uint8_t *edge1, const uint8_t *edge2, uint8_t *inlier) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int pr = pairOf[i];
    auto reproj2 = [](const double *S, const float *K, const float *X, const float *kp) -> float {
        const DQuat q{S[0], S[1], S[2], S[3]};
        const D3 r = quat_rotate(q, D3{(double)X[0], (double)X[1], (double)X[2]});
        const double s = S[7];
        const double px = s * r.x + S[4], py = s * r.y + S[5], pz = s * r.z + S[6];
        const float u = (float)((double)K[0] * px / pz + (double)K[2]), v = (float)((double)K[1] * py / pz + (double)K[3]);
        const float dx = kp[0] - u, dy = kp[1] - v;
        return dx * dx + dy * dy;
    };
    const float err1 = reproj2(Sc1w2 + (size_t)pr * 8, K1, X2 + (size_t)i * 3, kp1 + (size_t)i * 2);   // map-2 point into key-frame 1
    const float err2 = reproj2(Sc2w1 + (size_t)pr * 8, K2, X1 + (size_t)i * 3, kp2 + (size_t)i * 2);   // map-1 point into key-frame 2
    const bool ok1 = (double)err1 < 2 * 9.210 * (double)sigma1[i] || edge2[i];
    const bool ok2 = (double)err2 < 2 * 9.210 * (double)sigma2[i] || edge1[i];
    inlier[i] = ok1 && ok2;
}


// ==================================================================================================================
// Sim3Solver::iterate (R/lib_src/Sim3Solver.cc:159-404): the hypotheses of one block of RANSAC iterations, one workgroup each.
// Lane 0 forms the hypothesis from its three correspondences (ComputeSim3 :437-540: Horn's closed form; float arithmetic as upstream
// up to the 4x4 matrix N, whose dominant eigenvector comes from a cyclic Jacobi iteration in double — upstream calls Eigen's general
// EigenSolver<Matrix4f>, which is not in the tree: "parity unpinned"), then the workgroup runs CheckInliers (:542-562)
// over all correspondences and, for the rumination overload (:292-404), ComputeInliersNum (:564-664) under
// gSw1w2 = gSc1w^-1 * gSc1c2 * gSc2w (:344-347) over every matched key-point pair of every key-frame pair.
// ==================================================================================================================
struct RansacArgs {
    int n, nHyp, fixScale;
    const float *X1, *X2, *thr1, *thr2, *K1, *K2;
    const int32_t *tri;
    float *T12; int32_t *nIn; uint8_t *inl;
    int nPairs, total;                                            // score set (total == 0: none)
    const int32_t *pairOf; const double *Sc1w1, *Sc2w2, *Skf;
    const float *sK1, *sK2, *sX1, *sX2, *kp1, *kp2, *sg1, *sg2; const uint8_t *e1, *e2;
    int32_t *pairCnt; double *comp;
};

// dominant eigenvector (largest eigenvalue) of a symmetric 4x4 matrix: cyclic Jacobi rotations
__device__ inline void sym4_dominant_eigenvector(double a[4][4], double q[4]) {
    double v[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0;
        for (int i = 0; i < 4; i++) for (int j = i + 1; j < 4; j++) off += a[i][j] * a[i][j];
        if (off < 1e-300) break;
        for (int p = 0; p < 3; p++)
            for (int r = p + 1; r < 4; r++) {
                if (a[p][r] == 0) continue;
                const double theta = (a[r][r] - a[p][p]) / (2 * a[p][r]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                const double c = 1 / sqrt(t * t + 1), sn = t * c;
                for (int k = 0; k < 4; k++) { const double x = a[k][p], y = a[k][r]; a[k][p] = c * x - sn * y; a[k][r] = sn * x + c * y; }
                for (int k = 0; k < 4; k++) { const double x = a[p][k], y = a[r][k]; a[p][k] = c * x - sn * y; a[r][k] = sn * x + c * y; }
                for (int k = 0; k < 4; k++) { const double x = v[k][p], y = v[k][r]; v[k][p] = c * x - sn * y; v[k][r] = sn * x + c * y; }
            }
    }
    int best = 0;
    for (int i = 1; i < 4; i++) if (a[i][i] > a[best][best]) best = i;
    for (int k = 0; k < 4; k++) q[k] = v[k][best];
}

__global__ __launch_bounds__(256) void k_sim3_ransac(RansacArgs A) {
    __shared__ float sT12[12], sT21[12];      // rows of [sR | t]
    __shared__ int sCnt;
    __shared__ DSim3 sSw1w2;
    const int h = blockIdx.x, tid = threadIdx.x;
    float *Tout = A.T12 + (size_t)h * 16;
    if (tid == 0) {
        sCnt = 0;
        float P1[3][3], P2[3][3];                                  // column i = correspondence i (:185-188)
        for (int i = 0; i < 3; i++) {
            const int idx = A.tri[h * 3 + i];
            for (int r = 0; r < 3; r++) { P1[r][i] = A.X1[idx * 3 + r]; P2[r][i] = A.X2[idx * 3 + r]; }
        }
        float O1[3], O2[3], Pr1[3][3], Pr2[3][3];                  // ComputeCentroid :430-435
        for (int r = 0; r < 3; r++) {
            O1[r] = ((P1[r][0] + P1[r][1]) + P1[r][2]) / 3.f;
            O2[r] = ((P2[r][0] + P2[r][1]) + P2[r][2]) / 3.f;
            for (int i = 0; i < 3; i++) { Pr1[r][i] = P1[r][i] - O1[r]; Pr2[r][i] = P2[r][i] - O2[r]; }
        }
        float M[3][3];                                             // Pr2 * Pr1^T :453
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) M[r][c] = (Pr2[r][0] * Pr1[c][0] + Pr2[r][1] * Pr1[c][1]) + Pr2[r][2] * Pr1[c][2];
        const double N11 = M[0][0] + M[1][1] + M[2][2], N12 = M[1][2] - M[2][1], N13 = M[2][0] - M[0][2], N14 = M[0][1] - M[1][0],
                     N22 = M[0][0] - M[1][1] - M[2][2], N23 = M[0][1] + M[1][0], N24 = M[2][0] + M[0][2], N33 = -M[0][0] + M[1][1] - M[2][2],
                     N34 = M[1][2] + M[2][1], N44 = -M[0][0] - M[1][1] + M[2][2];
        // upstream stores N in a Matrix4f: the solver sees the float-rounded entries
        double Nm[4][4] = {{(float)N11, (float)N12, (float)N13, (float)N14}, {(float)N12, (float)N22, (float)N23, (float)N24},
                           {(float)N13, (float)N23, (float)N33, (float)N34}, {(float)N14, (float)N24, (float)N34, (float)N44}};
        double q[4];
        sym4_dominant_eigenvector(Nm, q);
        const float e0 = (float)q[0];
        float vec[3] = {(float)q[1], (float)q[2], (float)q[3]};
        const float nrm = sqrtf((vec[0] * vec[0] + vec[1] * vec[1]) + vec[2] * vec[2]);
        int valid = !(vec[0] == 0 && vec[1] == 0 && vec[2] == 0);   // :493-494 upstream keeps the previous iteration's transform
        float R[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, s12 = 1.f, t12[3] = {0, 0, 0};
        if (valid) {
            const double ang = atan2((double)nrm, (double)e0);
            const float f = (float)(2 * ang);
            for (int k = 0; k < 3; k++) vec[k] = vec[k] * f / nrm;   // angle-axis; the quaternion angle is the half
            // Sophus::SO3f::exp(vec).matrix()
            const float th2 = (vec[0] * vec[0] + vec[1] * vec[1]) + vec[2] * vec[2], th = sqrtf(th2), half = 0.5f * th;
            float im, re;
            if (th < 1e-5f) { const float th4 = th2 * th2; im = 0.5f - (1.f / 48.f) * th2 + (1.f / 3840.f) * th4; re = 1.f - (1.f / 8.f) * th2 + (1.f / 384.f) * th4; }
            else { im = sinf(half) / th; re = cosf(half); }
            const float qx = im * vec[0], qy = im * vec[1], qz = im * vec[2], qw = re;
            const float tx = 2 * qx, ty = 2 * qy, tz = 2 * qz, twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx,
                        tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
            R[0][0] = 1 - (tyy + tzz); R[0][1] = txy - twz; R[0][2] = txz + twy;
            R[1][0] = txy + twz; R[1][1] = 1 - (txx + tzz); R[1][2] = tyz - twx;
            R[2][0] = txz - twy; R[2][1] = tyz + twx; R[2][2] = 1 - (txx + tyy);
            if (!A.fixScale) {                                      // :503-520
                double nom = 0, den = 0;
                float P3[3][3];
                for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) P3[r][c] = (R[r][0] * Pr2[0][c] + R[r][1] * Pr2[1][c]) + R[r][2] * Pr2[2][c];
                float fn = 0, fd = 0;                               // Eigen's float array sums, column-major order
                for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) { fn += Pr1[r][c] * P3[r][c]; fd += P3[r][c] * P3[r][c]; }
                nom = fn; den = fd;
                s12 = (float)(nom / den);
            }
            for (int r = 0; r < 3; r++) {                           // mt12i = O1 - ms12i * mR12i * O2
                const float ro = ((s12 * R[r][0]) * O2[0] + (s12 * R[r][1]) * O2[1]) + (s12 * R[r][2]) * O2[2];
                t12[r] = O1[r] - ro;
            }
        }
        const float sinv = (float)(1.0 / s12);
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) { sT12[r * 4 + c] = s12 * R[r][c]; sT21[r * 4 + c] = sinv * R[c][r]; }
            sT12[r * 4 + 3] = t12[r];
        }
        for (int r = 0; r < 3; r++) sT21[r * 4 + 3] = -((sT21[r * 4 + 0] * t12[0] + sT21[r * 4 + 1] * t12[1]) + sT21[r * 4 + 2] * t12[2]);
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Tout[r * 3 + c] = R[r][c];
        Tout[9] = t12[0]; Tout[10] = t12[1]; Tout[11] = t12[2]; Tout[12] = s12; Tout[13] = (float)valid; Tout[14] = 0; Tout[15] = 0;
        if (A.total > 0) {                                          // :338-347
            double Rd[3][3];
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Rd[r][c] = (double)R[r][c];
            const DSim3 Sc1c2{quat_from_matrix(Rd), {(double)t12[0], (double)t12[1], (double)t12[2]}, (double)s12};
            sSw1w2 = sim3_mul(sim3_mul(sim3_inverse(sim3_from8(A.Skf)), Sc1c2), sim3_from8(A.Skf + 8));
        }
    }
    __syncthreads();
    // CheckInliers :542-562 (float, as upstream)
    const float fx1 = A.K1[0], fy1 = A.K1[1], cx1 = A.K1[2], cy1 = A.K1[3], fx2 = A.K2[0], fy2 = A.K2[1], cx2 = A.K2[2], cy2 = A.K2[3];
    int mine = 0;
    for (int i = tid; i < A.n; i += 256) {
        const float *a = A.X1 + (size_t)i * 3, *b = A.X2 + (size_t)i * 3;
        const float u1 = fx1 * a[0] / a[2] + cx1, v1 = fy1 * a[1] / a[2] + cy1;            // mvP1im1
        const float u2 = fx2 * b[0] / b[2] + cx2, v2 = fy2 * b[1] / b[2] + cy2;            // mvP2im2
        float p[3], r[3];
        for (int k = 0; k < 3; k++) {
            p[k] = ((sT12[k * 4] * b[0] + sT12[k * 4 + 1] * b[1]) + sT12[k * 4 + 2] * b[2]) + sT12[k * 4 + 3];   // point 2 in camera 1
            r[k] = ((sT21[k * 4] * a[0] + sT21[k * 4 + 1] * a[1]) + sT21[k * 4 + 2] * a[2]) + sT21[k * 4 + 3];   // point 1 in camera 2
        }
        const float d1x = u1 - (fx1 * p[0] / p[2] + cx1), d1y = v1 - (fy1 * p[1] / p[2] + cy1);
        const float d2x = (fx2 * r[0] / r[2] + cx2) - u2, d2y = (fy2 * r[1] / r[2] + cy2) - v2;
        const float err1 = d1x * d1x + d1y * d1y, err2 = d2x * d2x + d2y * d2y;
        const bool in = err1 < A.thr1[i] && err2 < A.thr2[i];
        if (A.inl) A.inl[(size_t)h * A.n + i] = in;
        mine += in;
    }
    if (mine) atomicAdd(&sCnt, mine);
    __syncthreads();
    if (tid == 0) A.nIn[h] = sCnt;
    if (A.total <= 0) return;
    // ComputeInliersNum :564-664 under this hypothesis
    double *comp = A.comp + (size_t)h * A.nPairs * 16;
    int32_t *cnt = A.pairCnt + (size_t)h * A.nPairs;
    for (int p = tid; p < A.nPairs; p += 256) {
        sim3_to8(sim3_mul(sim3_from8(A.Sc1w1 + (size_t)p * 8), sSw1w2), comp + (size_t)p * 16);                       // gSc1w2 :621
        sim3_to8(sim3_mul(sim3_from8(A.Sc2w2 + (size_t)p * 8), sim3_inverse(sSw1w2)), comp + (size_t)p * 16 + 8);     // gSc2w1 :620
        cnt[p] = 0;
    }
    __threadfence_block();
    __syncthreads();
    auto reproj2 = [](const double *S, const float *K, const float *X, const float *kp) -> float {
        const DQuat q{S[0], S[1], S[2], S[3]};
        const D3 r = quat_rotate(q, D3{(double)X[0], (double)X[1], (double)X[2]});
        const double s = S[7];
        const double px = s * r.x + S[4], py = s * r.y + S[5], pz = s * r.z + S[6];
        const float u = (float)((double)K[0] * px / pz + (double)K[2]), v = (float)((double)K[1] * py / pz + (double)K[3]);
        const float dx = kp[0] - u, dy = kp[1] - v;
        return dx * dx + dy * dy;
    };
    for (int i = tid; i < A.total; i += 256) {
        const int pr = A.pairOf[i];
        const float err1 = reproj2(comp + (size_t)pr * 16, A.sK1, A.sX2 + (size_t)i * 3, A.kp1 + (size_t)i * 2);
        const float err2 = reproj2(comp + (size_t)pr * 16 + 8, A.sK2, A.sX1 + (size_t)i * 3, A.kp2 + (size_t)i * 2);
        const bool ok1 = (double)err1 < 2 * 9.210 * (double)A.sg1[i] || A.e2[i];
        const bool ok2 = (double)err2 < 2 * 9.210 * (double)A.sg2[i] || A.e1[i];
        if (ok1 && ok2) atomicAdd(&cnt[pr], 1);
    }
}


// ==================================================================================================================
// OptimizeSim3 / OptimizeCloudSim3 (R/lib_src/Optimizer.cc:1920-2167, :2169-2471): one Sim3 vertex, fixed points, two reprojection
// edges per correspondence, numeric Jacobians (G/core/base_binary_edge.hpp:131-203, delta 1e-9, through VertexSim3Expmap::oplusImpl).
// One 256-thread workgroup runs both optimize() calls.  The transform an edge applies depends only on its key-frame pair and on the
// perturbation (gSc1w * est' * gSc2w^-1 and gSc2w * est'^-1 * gSc1w^-1, OptimizableTypes.h:242,285): the 15 (base, +-delta per
// dimension) x 2 composites per pair are formed once per linearisation and every correspondence evaluates 30 map + project against them.
// ==================================================================================================================
struct Sim3Args {
    int n, nPairs, world, fixScale, robustFirst;
    float th2;
    const int32_t *pairOf;
    const double *Sc1w, *Sc2w, *Sin;
    const float *P1c, *P2c, *obs1, *obs2, *w1, *w2;
    const uint8_t *skip12, *skip21;
    const float *K1, *K2;
    double *Sout; int32_t *res; uint8_t *status;          // results
    double *comp, *chi12, *chi21; uint8_t *on12, *on21;   // scratch
};

__global__ __launch_bounds__(256) void k_sim3_opt(Sim3Args A) {
    __shared__ double red[5 * 64];
    const int tid = threadIdx.x, n = A.n, np = A.world ? A.nPairs : 1;
    const DCam cam1{A.K1[0], A.K1[1], A.K1[2], A.K1[3]}, cam2{A.K2[0], A.K2[1], A.K2[2], A.K2[3]};
    const double delta = (double)sqrtf(A.th2), dsqr = delta * delta, th2 = (double)A.th2;
    DSim3 est = sim3_from8(A.Sin);
    bool robust = A.robustFirst != 0;
    for (int i = tid; i < n; i += 256) { A.on12[i] = !(A.skip12 && A.skip12[i]); A.on21[i] = !(A.skip21 && A.skip21[i]); A.status[i] = 0; }

    auto oplus = [&](const DSim3 &S, const double *upd) -> DSim3 {          // VertexSim3Expmap::oplusImpl
        double u[7];
#pragma unroll
        for (int k = 0; k < 7; k++) u[k] = upd[k];
        if (A.fixScale) u[6] = 0;
        return sim3_mul(sim3_exp(u), S);
    };
    auto fill = [&](const DSim3 &S0, bool full) {                            // composites of every pair: slot 0 base, 1 + 2d / 2 + 2d = +-delta in dimension d
        const int cnt = full ? 15 : 1;
        for (int idx = tid; idx < np * cnt; idx += 256) {
            const int p = idx / cnt, k = idx - p * cnt;
            DSim3 S = S0;
            if (k) {
                double add[7] = {0, 0, 0, 0, 0, 0, 0};
                const int d = (k - 1) >> 1;
                const double v = (k & 1) ? 1e-9 : -1e-9;
#pragma unroll
                for (int q = 0; q < 7; q++) if (q == d) add[q] = v;
                S = oplus(S0, add);
            }
            DSim3 F = S, I = sim3_inverse(S);
            if (A.world) {
                const DSim3 a = sim3_from8(A.Sc1w + (size_t)p * 8), b = sim3_from8(A.Sc2w + (size_t)p * 8);
                F = sim3_mul(sim3_mul(a, S), sim3_inverse(b));
                I = sim3_mul(sim3_mul(b, sim3_inverse(S)), sim3_inverse(a));
            }
            sim3_to8(F, A.comp + ((size_t)p * 30 + k) * 8);
            sim3_to8(I, A.comp + ((size_t)p * 30 + 15 + k) * 8);
        }
        __threadfence_block();
        __syncthreads();
    };
    auto err12 = [&](int i, int p, int k, double &e0, double &e1) {
        const D3 pc = sim3_map(sim3_from8(A.comp + ((size_t)p * 30 + k) * 8), D3{(double)A.P2c[3 * i], (double)A.P2c[3 * i + 1], (double)A.P2c[3 * i + 2]});
        double u, v;
        cam_project(cam1, pc, u, v);
        e0 = (double)A.obs1[2 * i] - u; e1 = (double)A.obs1[2 * i + 1] - v;
    };
    auto err21 = [&](int i, int p, int k, double &e0, double &e1) {
        const D3 pc = sim3_map(sim3_from8(A.comp + ((size_t)p * 30 + 15 + k) * 8), D3{(double)A.P1c[3 * i], (double)A.P1c[3 * i + 1], (double)A.P1c[3 * i + 2]});
        double u, v;
        cam_project(cam2, pc, u, v);
        e0 = (double)A.obs2[2 * i] - u; e1 = (double)A.obs2[2 * i + 1] - v;
    };
    auto robust_chi2 = [&](const DSim3 &S) -> double {                       // computeActiveErrors + activeRobustChi2
        fill(S, false);
        double acc[1] = {0};
        for (int i = tid; i < n; i += 256) {
            const int p = A.pairOf ? A.pairOf[i] : 0;
            if (A.on12[i]) {
                double e0, e1; err12(i, p, 0, e0, e1);
                const double w = (double)A.w1[i], c = e0 * w * e0 + e1 * w * e1;
                A.chi12[i] = c;
                double r0 = c, r1 = 1;
                if (robust) huber(c, delta, dsqr, r0, r1);
                acc[0] += r0;
            }
            if (A.on21[i]) {
                double e0, e1; err21(i, p, 0, e0, e1);
                const double w = (double)A.w2[i], c = e0 * w * e0 + e1 * w * e1;
                A.chi21[i] = c;
                double r0 = c, r1 = 1;
                if (robust) huber(c, delta, dsqr, r0, r1);
                acc[0] += r0;
            }
        }
        block_sum<1>(acc, red);
        return acc[0];
    };
    auto lm = [&](int maxIt) {                                               // optimization_algorithm_levenberg.cpp:61-169
        double lambda = -1, ni = 2;
        int nBadIt = 0;
        for (int itl = 0; itl < maxIt; itl++) {
            double hb[36];                                                   // 28 upper entries of H, 7 of b, robust chi2
#pragma unroll
            for (int k = 0; k < 36; k++) hb[k] = 0;
            fill(est, true);
            for (int i = tid; i < n; i += 256) {
                const int p = A.pairOf ? A.pairOf[i] : 0;
#pragma unroll
                for (int side = 0; side < 2; side++) {
                    if (!(side ? A.on21[i] : A.on12[i])) continue;
                    double e0, e1, J0[7], J1[7];
                    if (side) err21(i, p, 0, e0, e1); else err12(i, p, 0, e0, e1);
#pragma unroll
                    for (int d = 0; d < 7; d++) {
                        double p0, p1, m0, m1;
                        if (side) { err21(i, p, 1 + 2 * d, p0, p1); err21(i, p, 2 + 2 * d, m0, m1); }
                        else { err12(i, p, 1 + 2 * d, p0, p1); err12(i, p, 2 + 2 * d, m0, m1); }
                        J0[d] = 5e8 * (p0 - m0); J1[d] = 5e8 * (p1 - m1);       // scalar = 1 / (2 delta)
                    }
                    const double w = (double)(side ? A.w2[i] : A.w1[i]), c = e0 * w * e0 + e1 * w * e1;
                    if (side) A.chi21[i] = c; else A.chi12[i] = c;
                    double r0 = c, r1 = 1;
                    if (robust) huber(c, delta, dsqr, r0, r1);
                    hb[35] += r0;
                    const double rw = r1 * w;
                    int q = 0;
#pragma unroll
                    for (int a = 0; a < 7; a++) {
#pragma unroll
                        for (int c2 = a; c2 < 7; c2++) hb[q++] += rw * (J0[a] * J0[c2] + J1[a] * J1[c2]);
                    }
#pragma unroll
                    for (int a = 0; a < 7; a++) hb[28 + a] -= r1 * (J0[a] * w * e0 + J1[a] * w * e1);
                }
            }
            block_sum_butterfly<36>(hb, red);
            double currentChi = hb[35];
            const double iniChi = currentChi;
            if (itl == 0) {
                double m = 0;
                int q = 0;
                for (int a = 0; a < 7; a++) { m = fmax(fabs(hb[q]), m); q += 7 - a; }
                lambda = 1e-5 * m; ni = 2; nBadIt = 0;
            }
            double rho = 0;
            int qmax = 0;
            do {
                const DSim3 saved = est;
                double x[7];
                const bool ok2 = chol_solve_packed<7>(hb, lambda, hb + 28, x);
                if (ok2) est = oplus(est, x);
                double tempChi = robust_chi2(est);
                if (!ok2) tempChi = DBL_MAX;
                rho = currentChi - tempChi;
                double scale = 0;
                if (ok2) for (int j = 0; j < 7; j++) scale += x[j] * (lambda * x[j] + hb[28 + j]);
                scale += 1e-3;
                rho /= scale;
                if (rho > 0 && isfinite(tempChi)) {
                    const double tr = 2 * rho - 1;
                    double alpha = 1. - tr * tr * tr;
                    alpha = fmin(alpha, 2. / 3.);
                    lambda *= fmax(1. / 3., alpha);
                    ni = 2;
                    currentChi = tempChi;
                } else {
                    lambda *= ni;
                    ni *= 2;
                    est = saved;
                }
                qmax++;
            } while (rho < 0 && qmax < 10);
            if (qmax == 10 || rho == 0) break;
            if ((iniChi - currentChi) * 1e3 < iniChi) nBadIt++; else nBadIt = 0;
            if (nBadIt >= 3) break;
        }
    };

    __syncthreads();
    if (n > 0) lm(5);                                                        // optimizer.optimize(5)
    if (tid == 0) sim3_to8(est, A.Sout);                                     // OptimizeCloudSim3 publishes this estimate already (:2397)
    double bad[1] = {0};
    for (int i = tid; i < n; i += 256) {                                     // :2110 / :2407: chi2() of the errors the last computeActiveErrors() left
        if ((A.on12[i] && A.chi12[i] > th2) || (A.on21[i] && A.chi21[i] > th2)) { A.status[i] = 1; A.on12[i] = 0; A.on21[i] = 0; bad[0] += 1; }
    }
    block_sum<1>(bad, red);
    const int nBad = (int)bad[0];
    robust = false;                                                          // setRobustKernel(0)
    if (n - nBad < 10) {
        if (tid == 0) { A.res[0] = 0; A.res[1] = nBad; A.res[2] = 1; }
        return;
    }
    lm(nBad > 0 ? 10 : 5);
    fill(est, false);
    double in[1] = {0};
    for (int i = tid; i < n; i += 256) {
        if (A.status[i] == 1) continue;
        if (!A.on12[i] || !A.on21[i]) { A.status[i] = 3; continue; }         // :2450-2451
        const int p = A.pairOf ? A.pairOf[i] : 0;
        double a0, a1, b0, b1;
        err12(i, p, 0, a0, a1); err21(i, p, 0, b0, b1);
        const double w1 = (double)A.w1[i], w2 = (double)A.w2[i];
        if (a0 * w1 * a0 + a1 * w1 * a1 > th2 || b0 * w2 * b0 + b1 * w2 * b1 > th2) A.status[i] = 2; else in[0] += 1;
    }
    block_sum<1>(in, red);
    if (tid == 0) { sim3_to8(est, A.Sout); A.res[0] = (int)in[0]; A.res[1] = nBad; A.res[2] = 0; }
}

__device__ __forceinline__ double wave_allreduce_max(double v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = fmax(v, __shfl_xor(v, d));
    return v;
}

#include "ba_windows.inc"

}  // namespace rumi

using namespace rumi;

// ================================================ host side =======================================================
struct RumiOptimizer {
    int device = 0;
    int maxPoseEdges = 0, maxPoseBatch = 0, maxKF = 0, maxMP = 0, maxE = 0;
    // pose optimisation
    uint8_t *dActive = nullptr, *dEOff = nullptr;       // pose inputs / outputs live in the transfer blocks below
    double *dLastChi2 = nullptr;
    // BA
    double *dT[2] = {nullptr, nullptr}, *dX[2] = {nullptr, nullptr};   // graph arrays are read in place from the upload mirror (dBa)
    double *dHll = nullptr, *dBl = nullptr, *dHpl = nullptr, *dPanel = nullptr, *dHpp = nullptr, *dBp = nullptr, *dDinv = nullptr,
           *dS = nullptr, *dBs = nullptr, *dXv = nullptr, *dChi = nullptr, *dScal = nullptr, *dAglob = nullptr, *dYt = nullptr, *dG = nullptr, *dLp = nullptr;
    int npCap = 0;
    uint8_t *dErase = nullptr;
    double *dW = nullptr;            // H_pl L per edge, allocated by the first large-window call
    int32_t *dColOf = nullptr;       // column block of every edge's key-frame (-1 fixed), same
    int32_t *dPairs = nullptr; size_t pairCap = 0, pairOff = 0;   // Schur block descriptors + observation pairs of the large-window path
    double *hScal = nullptr;         // fine-grained pinned: [0..7] the trial's scalars, [8] sequence number of the last publication (k_ba_publish)
    double *dhScal = nullptr;        // the same memory as the device sees it
    // device-side LM control (LmState on the device; its mirror and the forwarded stop flag in fine-grained pinned memory)
    rumi::LmState *dLm = nullptr;
    rumi::LmMirror *hLm = nullptr, *dhLm = nullptr;
    int32_t *hStop = nullptr, *dhStop = nullptr;
    rumi::LmState *hLmInit = nullptr;    // pinned staging of the initial state
    unsigned long long lmSeq = 0;
    unsigned long long pubSeq = 0;
    // window-batched local BA (ba_windows.inc): per-arena extras, allocated on first use; the window table lives in the handle that runs the batch
    double *dGpart = nullptr, *dGw = nullptr, *dChiPart = nullptr, *dSclPart = nullptr;
    uint8_t *dWinSmall = nullptr, *dSorted = nullptr; size_t sortedBytes = 0;
    rumi::LmCtl *dLmCtl = nullptr, *hLmCtl = nullptr;
    rumi::WinMirror *hWm = nullptr, *dhWm = nullptr;
    rumi::BAWin *hWinTab = nullptr, *dWinTab = nullptr;
    unsigned bawRun = 0;
    uint8_t *dGroupOut = nullptr, *hGroupOut = nullptr; size_t groupOutCap = 0;     // results of a launch group, gathered for one copy back
    hipStream_t stream = nullptr;    // bundle adjustments of this handle (created non-blocking)
    std::vector<RumiOptimizer *> workers;   // rumi_local_ba_batch: one child handle per worker thread, created on first use
    int maxKFc = 0, maxMPc = 0, maxEc = 0;   // creation arguments (children are created alike)
    uint8_t *hPose = nullptr, *hPoseOut = nullptr, *dPoseIn = nullptr, *dPoseOut = nullptr;   // PoseOptimization transfer blocks
    uint8_t *hBa = nullptr, *dBa = nullptr, *dBaOut = nullptr; size_t baStageCap = 0;            // bundle-adjustment transfer blocks
    std::vector<int32_t> hFill;                                                                  // counting-sort cursors of ba_run
    float stageMs[8] = {0};
    hipEvent_t ev[2] = {nullptr, nullptr};
    // opt-in per-kernel timing of the bundle adjustment (rumi_opt_set_profiling): event pairs around the pose-block Gram product, the Schur
    // SYRK and the reduced solve of every trial, summed after the trial's own synchronisation
    bool profiling = false;
    hipEvent_t evK[6] = {nullptr};
    float kernelMs[4] = {0};          // hpp, syrk, solve (ms over the call), trials
};

template <class T> static int oalloc(T **p, size_t n) {
    *p = nullptr;
    HIP_TRY(hipMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(T)));
    return RUMI_OK;
}

extern "C" void rumi_opt_destroy(RumiOptimizer *o) {
    if (!o) return;
    (void)hipSetDevice(o->device);
    for (RumiOptimizer *w : o->workers) rumi_opt_destroy(w);
    o->workers.clear();
    if (o->stream) (void)hipStreamDestroy(o->stream);
    void *p[] = {o->dActive, o->dLastChi2, o->dT[0], o->dT[1], o->dX[0],
                 o->dX[1], o->dHll, o->dBl, o->dHpl, o->dPanel, o->dHpp, o->dBp, o->dDinv, o->dS, o->dBs, o->dXv, o->dChi, o->dScal,
                 o->dAglob, o->dErase, o->dEOff, o->dYt, o->dG, o->dLp, o->dW, o->dColOf};
    for (void *q : p) if (q) (void)hipFree(q);
    if (o->dPairs) (void)hipFree(o->dPairs);
    { void *q[] = {o->dGpart, o->dGw, o->dChiPart, o->dSclPart, o->dWinSmall, o->dLmCtl, o->dWinTab, o->dSorted}; for (void *x : q) if (x) (void)hipFree(x); }
    if (o->hLmCtl) (void)hipHostFree(o->hLmCtl);
    if (o->hWm) (void)hipHostFree((void *)o->hWm);
    if (o->hWinTab) (void)hipHostFree(o->hWinTab);
    if (o->dGroupOut) (void)hipFree(o->dGroupOut);
    if (o->hGroupOut) (void)hipHostFree(o->hGroupOut);
    if (o->hScal) (void)hipHostFree(o->hScal);
    if (o->hLm) (void)hipHostFree(o->hLm);
    if (o->hStop) (void)hipHostFree(o->hStop);
    if (o->hLmInit) (void)hipHostFree(o->hLmInit);
    if (o->dLm) (void)hipFree(o->dLm);
    if (o->hPose) (void)hipHostFree(o->hPose);
    if (o->hBa) (void)hipHostFree(o->hBa);
    if (o->dBa) (void)hipFree(o->dBa);
    if (o->dBaOut) (void)hipFree(o->dBaOut);
    if (o->hPoseOut) (void)hipHostFree(o->hPoseOut);
    if (o->dPoseIn) (void)hipFree(o->dPoseIn);
    if (o->dPoseOut) (void)hipFree(o->dPoseOut);
    for (auto &e : o->ev) if (e) (void)hipEventDestroy(e);
    delete o;
}

extern "C" int rumi_opt_create(int32_t max_pose_edges, int32_t max_pose_batch, int32_t max_kf, int32_t max_mp, int32_t max_edges,
                               int32_t device, RumiOptimizer **out) {
    if (!out) return RUMI_E_INVALID;
    *out = nullptr;
    if (max_pose_edges < 1 || max_pose_batch < 1 || max_kf < 1 || max_mp < 1 || max_edges < 1) { g_lastError = "rumi_opt_create: sizes must be >= 1"; return RUMI_E_INVALID; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_lastError = "no HIP device visible: librumi_hip has no CPU fallback"; return RUMI_E_NO_DEVICE; }
    RumiOptimizer *o = new RumiOptimizer();
    if (device >= 0) o->device = device; else if (hipGetDevice(&o->device) != hipSuccess) o->device = 0;
    if (hipSetDevice(o->device) != hipSuccess) { delete o; return RUMI_E_NO_DEVICE; }
    o->maxPoseEdges = max_pose_edges; o->maxPoseBatch = max_pose_batch; o->maxKF = max_kf; o->maxMP = max_mp; o->maxE = max_edges;
    if (hipStreamCreateWithFlags(&o->stream, hipStreamNonBlocking) != hipSuccess) { delete o; return RUMI_E_NO_DEVICE; }
    const size_t PE = max_pose_edges, PB = max_pose_batch, K = max_kf, M = max_mp, E = max_edges, N = 6 * K;
    int rc;
#define TRYA(x) if ((rc = (x)) != RUMI_OK) { rumi_opt_destroy(o); return rc; }
    TRYA(oalloc(&o->dActive, PE)); TRYA(oalloc(&o->dLastChi2, PE));
    for (int i = 0; i < 2; i++) { TRYA(oalloc(&o->dT[i], K * 8)); TRYA(oalloc(&o->dX[i], M * 3)); }
    TRYA(oalloc(&o->dHll, M * 9)); TRYA(oalloc(&o->dBl, M * 3)); TRYA(oalloc(&o->dHpl, E * 18)); TRYA(oalloc(&o->dPanel, E * 16 + 64));
    TRYA(oalloc(&o->dHpp, K * 36)); TRYA(oalloc(&o->dBp, N)); TRYA(oalloc(&o->dDinv, M * 9)); TRYA(oalloc(&o->dS, N * N));
    TRYA(oalloc(&o->dBs, N)); TRYA(oalloc(&o->dXv, N + M * 3)); TRYA(oalloc(&o->dChi, E)); TRYA(oalloc(&o->dScal, 8));
    o->npCap = (int)std::min<size_t>((N + 1 + 15) / 16 * 16, 256);
    TRYA(oalloc(&o->dAglob, (N + 2) * (N + 2) + 2 * N)); TRYA(oalloc(&o->dErase, E)); TRYA(oalloc(&o->dEOff, E));
    TRYA(oalloc(&o->dYt, 3 * M * (size_t)o->npCap)); TRYA(oalloc(&o->dG, (size_t)o->npCap * o->npCap)); TRYA(oalloc(&o->dLp, M * 6));
#undef TRYA
    if (hipHostMalloc((void **)&o->hScal, 16 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer((void **)&o->dhScal, o->hScal, 0) != hipSuccess) { rumi_opt_destroy(o); return RUMI_E_NO_DEVICE; }
    std::memset(o->hScal, 0, 16 * sizeof(double));
    if (hipHostMalloc((void **)&o->hLm, sizeof(rumi::LmMirror), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer((void **)&o->dhLm, o->hLm, 0) != hipSuccess ||
        hipHostMalloc((void **)&o->hStop, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer((void **)&o->dhStop, o->hStop, 0) != hipSuccess ||
        hipHostMalloc((void **)&o->hLmInit, sizeof(rumi::LmState), hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void **)&o->dLm, sizeof(rumi::LmState)) != hipSuccess) { rumi_opt_destroy(o); return RUMI_E_NO_DEVICE; }
    std::memset((void *)o->hLm, 0, sizeof(rumi::LmMirror)); *o->hStop = 0;
    o->baStageCap = E * 48 + M * 32 + K * 80 + 1024;
    if (hipHostMalloc((void **)&o->hBa, o->baStageCap, hipHostMallocDefault) != hipSuccess || hipMalloc((void **)&o->dBa, o->baStageCap) != hipSuccess ||
        hipMalloc((void **)&o->dBaOut, o->baStageCap) != hipSuccess) {
        rumi_opt_destroy(o);
        return RUMI_E_NO_DEVICE;
    }
    {
        const size_t inCap = (PB + 1) * 4 + 16 + PB * 28 + PE * 24 + 256, outCap = PB * 32 + PE + 256;
        if (hipHostMalloc((void **)&o->hPose, inCap, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc((void **)&o->hPoseOut, outCap, hipHostMallocDefault) != hipSuccess ||
            hipMalloc((void **)&o->dPoseIn, inCap) != hipSuccess || hipMalloc((void **)&o->dPoseOut, outCap) != hipSuccess) {
            rumi_opt_destroy(o);
            return RUMI_E_NO_DEVICE;
        }
    }
    for (auto &e : o->ev) if (hipEventCreate(&e) != hipSuccess) { rumi_opt_destroy(o); return RUMI_E_NO_DEVICE; }
    for (auto &e : o->evK) if (hipEventCreate(&e) != hipSuccess) { rumi_opt_destroy(o); return RUMI_E_NO_DEVICE; }
    *out = o;
    return RUMI_OK;
}

extern "C" int rumi_opt_set_profiling(RumiOptimizer *o, int32_t on) {
    if (!o) return RUMI_E_INVALID;
    o->profiling = on != 0;
    return RUMI_OK;
}
extern "C" int rumi_opt_kernel_ms(RumiOptimizer *o, float ms[4]) {
    if (!o || !ms) return RUMI_E_INVALID;
    for (int i = 0; i < 4; i++) ms[i] = o->kernelMs[i];
    return RUMI_OK;
}

extern "C" int rumi_opt_stage_ms(RumiOptimizer *o, float ms[8]) {
    if (!o || !ms) return RUMI_E_INVALID;
    for (int i = 0; i < 8; i++) ms[i] = o->stageMs[i];
    return RUMI_OK;
}

namespace rumi {
int pose_opt_device(const int32_t *dStart, const float *dXw, const float *dObs, const float *dW, const float *dK4, const float *dTin, float *dTout,
                    uint8_t *dOutlier, int32_t *dNGood, uint8_t *dActive, double *dLastChi2, bool fitsLds, hipStream_t st) {
    const PoseArgs A{dStart, dXw, dObs, dW, dK4, dTin, dTout, dOutlier, dNGood, dActive, dLastChi2, 1};
    // the frame's size is known to the device only: both instantiations are launched, the one the size does not belong to returns at once
    hipLaunchKernelGGL((k_pose_opt<true, 256>), dim3(1), dim3(256), 0, st, A);
    if (!fitsLds) hipLaunchKernelGGL((k_pose_opt<false, 256>), dim3(1), dim3(256), 0, st, A);
    return hipGetLastError() == hipSuccess ? RUMI_OK : RUMI_E_NO_DEVICE;
}
}  // namespace rumi

extern "C" int rumi_pose_optimization_batch(RumiOptimizer *o, int32_t nbatch, const int32_t *start, const float *Xw, const float *obs,
                                            const float *inv_sigma2, const float *K4, float *Tcw7, uint8_t *outlier_out,
                                            int32_t *n_good_out) {
    if (!o || nbatch < 1 || !start || !K4 || !Tcw7 || !n_good_out) return RUMI_E_INVALID;
    const int total = start[nbatch];
    if (nbatch > o->maxPoseBatch || total > o->maxPoseEdges) { g_lastError = "pose optimisation: batch larger than the optimiser's arenas"; return RUMI_E_CAPACITY; }
    if (total > 0 && (!Xw || !obs || !inv_sigma2 || !outlier_out)) return RUMI_E_INVALID;
    HIP_TRY(hipSetDevice(o->device));
    // one pinned block up: [start | K4 | T | Xw | obs | w]; one block back: [nGood | T | outlier]
    auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t oStart = 0, oK = al(oStart + (size_t)(nbatch + 1) * 4), oT = al(oK + 16), oX = al(oT + (size_t)nbatch * 28),
                 oO = al(oX + (size_t)total * 12), oW = al(oO + (size_t)total * 8), inBytes = al(oW + (size_t)total * 4);
    const size_t rG = 0, rT = al(rG + (size_t)nbatch * 4), rO = al(rT + (size_t)nbatch * 28), outBytes = al(rO + (size_t)total);
    uint8_t *hs = o->hPose;
    std::memcpy(hs + oStart, start, (size_t)(nbatch + 1) * 4);
    std::memcpy(hs + oK, K4, 16);
    std::memcpy(hs + oT, Tcw7, (size_t)nbatch * 28);
    if (total > 0) {
        std::memcpy(hs + oX, Xw, (size_t)total * 12); std::memcpy(hs + oO, obs, (size_t)total * 8); std::memcpy(hs + oW, inv_sigma2, (size_t)total * 4);
    }
    HIP_TRY(hipMemcpyAsync(o->dPoseIn, hs, inBytes, hipMemcpyHostToDevice, nullptr));
    uint8_t *di = o->dPoseIn, *dout = o->dPoseOut;
    PoseArgs A{(const int32_t *)(di + oStart), (const float *)(di + oX), (const float *)(di + oO), (const float *)(di + oW), (const float *)(di + oK),
               (const float *)(di + oT), (float *)(dout + rT), dout + rO, (int32_t *)(dout + rG), o->dActive, o->dLastChi2, 1};
    bool anyBig = false, anySmall = false;
    for (int b = 0; b < nbatch; b++) { const int nb = start[b + 1] - start[b]; anyBig |= nb > kPoseLdsEdges; anySmall |= nb <= kPoseLdsEdges; }
    // 256 threads per frame: measured against 128 (216 us for one frame of 300 correspondences) and 512 (265 us) it is the fastest (194 us)
    if (anySmall) hipLaunchKernelGGL((k_pose_opt<true, 256>), dim3(nbatch), dim3(256), 0, nullptr, A);
    if (anyBig) hipLaunchKernelGGL((k_pose_opt<false, 256>), dim3(nbatch), dim3(256), 0, nullptr, A);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(o->hPoseOut, dout, outBytes, hipMemcpyDeviceToHost));
    std::memcpy(n_good_out, o->hPoseOut + rG, (size_t)nbatch * 4);
    std::memcpy(Tcw7, o->hPoseOut + rT, (size_t)nbatch * 28);           // early returns (< 3 correspondences) carry the input pose
    if (total > 0) std::memcpy(outlier_out, o->hPoseOut + rO, (size_t)total);
    return RUMI_OK;
}

extern "C" int rumi_pose_optimization(RumiOptimizer *o, const float *Xw, const float *obs, const float *inv_sigma2, int32_t n,
                                      const float *K4, float *Tcw7, uint8_t *outlier_out, int32_t *n_good_out) {
    if (n < 0) return RUMI_E_INVALID;
    const int32_t start[2] = {0, n};
    return rumi_pose_optimization_batch(o, 1, start, Xw, obs, inv_sigma2, K4, Tcw7, outlier_out, n_good_out);
}

#include "ba_windows_host.inc"

// mode 0: Optimizer::LocalBundleAdjustment(KeyFrame*, bool*, Map*, ...) — one optimize(10) with Huber(sqrt(5.991)).
// mode 1: Optimizer::LocalBundleAdjustment(KeyFrame *pMainKF, vpAdjustKF, vpFixedKF, bool*) (merge window, Optimizer.cc:3768-4183) —
//         optimize(5) with Huber(sqrt(5.99)); unless stopped: outlier edges to level 1, kernels off, initializeOptimization(0) +
//         optimize(10); the erase test reads every edge's stored error (level-1 edges: the one they had when they left).
// mode 2: Optimizer::BundleAdjustment(vpKFs, vpMP, nIterations, pbStopFlag, nLoopKF, bRobust) (Optimizer.cc:54-351, monocular edges) —
//         one optimize(nIterations), Huber(sqrt(5.99)) only if bRobust.
static int ba_run(RumiOptimizer *o, int mode, int32_t nKF, float *kf_pose7, const uint8_t *kf_fixed, int32_t nMP, float *mp_pos3,
                  int32_t nE, const int32_t *e_mp, const int32_t *e_kf, const float *e_obs, const float *e_inv_sigma2,
                  const float *K4, const volatile uint8_t *stop_flag, uint8_t *erase_out, int32_t *stats, int gbaIterations = 0, int gbaRobust = 1) {
    if (!o || nKF < 1 || nMP < 0 || nE < 0 || !kf_pose7 || !kf_fixed || !K4 || (nMP > 0 && !mp_pos3) ||
        (nE > 0 && (!e_mp || !e_kf || !e_obs || !e_inv_sigma2 || !erase_out)))
        return RUMI_E_INVALID;
    if (nKF > o->maxKF || nMP > o->maxMP || nE > o->maxE) { g_lastError = "local BA: problem larger than the optimiser's arenas"; return RUMI_E_CAPACITY; }
    // windows the tile solver takes (up to 29 optimised key-frames): the window-batched kernels, as a batch of one
    if (!o->profiling && !std::getenv("RUMI_BA_HOST_LM") && baw_eligible(nKF, kf_fixed, nMP, nE) && !(mode == 2 && (gbaIterations < 1 || (stop_flag && *stop_flag)))) {
        const BawArgs a{nKF, kf_pose7, kf_fixed, nMP, mp_pos3, nE, e_mp, e_kf, e_obs, e_inv_sigma2, K4, stop_flag, erase_out, stats};
        RumiOptimizer *arena = o;
        int32_t status = RUMI_OK;
        const int rc = baw_run(o, mode, 1, &a, &arena, gbaIterations, gbaRobust, &status);
        return rc != RUMI_OK ? rc : status;
    }
    if (stats) stats[0] = stats[1] = stats[2] = stats[3] = 0;
    int nFixed = 0;
    for (int k = 0; k < nKF; k++) nFixed += kf_fixed[k] ? 1 : 0;
    if (nFixed == 0 && mode == 0) { g_lastError = "LM-LBA: There are 0 fixed KF in the optimizations, LBA aborted"; return RUMI_E_INVALID; }   // Optimizer.cc:1057-1060
    if (mode != 2 && stop_flag && *stop_flag) { if (stats) stats[3] = 1; return RUMI_OK; }                                         // :1274-1276 / :3982-3984
    HIP_TRY(hipSetDevice(o->device));
    // everything of a bundle adjustment runs on the handle's own (non-blocking) stream: handles on different host threads overlap on the device
    // (rumi_local_ba_batch; Tracking / LocalMapping / LoopClosing threads with their thread-local arenas)
    hipStream_t st = o->stream;
    static const bool hostDbg = std::getenv("RUMI_HOSTDBG") != nullptr;
    auto now = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tA = now();

    // ---- one pinned block up, in two parts.  Part A is the caller's data (edges, measurements, initial state): converted first and already
    // on its way over PCIe while the host derives part B, the structure (g2o buildStructure: column blocks, edges by landmark, rows by
    // key-frame), written straight into the pinned block: two passes over the edges, no temporaries.  The kernels read both parts from the
    // device mirror; the initial state is copied on the device into the first of the two state buffers.
    int nOpt = 0;
    for (int k = 0; k < nKF; k++) nOpt += kf_fixed[k] ? 0 : 1;
    const int n = 6 * nOpt;
    auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t oEM = 0, oEK = al(oEM + (size_t)nE * 4), oOb = al(oEK + (size_t)nE * 4), oIn = al(oOb + (size_t)nE * 8),
                 oT = al(oIn + (size_t)nE * 4), oX = al(oT + (size_t)nKF * 64), partA = al(oX + (size_t)nMP * 24),
                 oPC = partA, oPS = al(oPC + (size_t)nKF * 4), oKR = al(oPS + (size_t)(nMP + 1) * 4), oPE = al(oKR + (size_t)(nOpt + 1) * 4),
                 oRS = al(oPE + (size_t)nE * 4), upBytes = al(oRS + (size_t)nE * 4);
    if (upBytes > o->baStageCap) { g_lastError = "local BA: upload block larger than the optimiser's arenas"; return RUMI_E_CAPACITY; }
    uint8_t *hs = o->hBa;
    int32_t *poseCol = reinterpret_cast<int32_t *>(hs + oPC), *ptStart = reinterpret_cast<int32_t *>(hs + oPS),
            *kfRowStart = reinterpret_cast<int32_t *>(hs + oKR), *ptEdge = reinterpret_cast<int32_t *>(hs + oPE),
            *rowSlot = reinterpret_cast<int32_t *>(hs + oRS);
    { int c = 0; for (int k = 0; k < nKF; k++) poseCol[k] = kf_fixed[k] ? -1 : c++; }
    // pass 1: validate, count edges per landmark and rows per optimised key-frame
    std::memset(ptStart, 0, (size_t)(nMP + 1) * 4);
    std::memset(kfRowStart, 0, (size_t)(nOpt + 1) * 4);
    {
        unsigned badIdx = 0;
        for (int e = 0; e < nE; e++) {
            const unsigned mp = (unsigned)e_mp[e], kf = (unsigned)e_kf[e];
            if (mp >= (unsigned)nMP || kf >= (unsigned)nKF) { badIdx = 1; break; }
            ptStart[mp + 1]++;
            const int c = poseCol[kf];
            if (c >= 0) kfRowStart[c + 1] += 2;
        }
        if (badIdx) { g_lastError = "local BA: edge index out of range"; return RUMI_E_INVALID; }
    }
    // part A
    if (nE > 0) {
        std::memcpy(hs + oEM, e_mp, (size_t)nE * 4); std::memcpy(hs + oEK, e_kf, (size_t)nE * 4);
        std::memcpy(hs + oOb, e_obs, (size_t)nE * 8); std::memcpy(hs + oIn, e_inv_sigma2, (size_t)nE * 4);
    }
    {
        double *T0 = reinterpret_cast<double *>(hs + oT), *X0 = reinterpret_cast<double *>(hs + oX);
        for (int k = 0; k < nKF; k++) {
            const DSE3 P = se3_from_float7(kf_pose7 + (size_t)k * 7);
            double *t = T0 + (size_t)k * 8;
            t[0] = P.r.x; t[1] = P.r.y; t[2] = P.r.z; t[3] = P.r.w; t[4] = P.t.x; t[5] = P.t.y; t[6] = P.t.z; t[7] = 0;
        }
        for (size_t i = 0; i < (size_t)nMP * 3; i++) X0[i] = mp_pos3[i];
    }
    const double tB = now();
    HIP_TRY(hipMemcpyAsync(o->dBa, hs, partA, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(o->dT[0], o->dBa + oT, (size_t)nKF * 64, hipMemcpyDeviceToDevice, st));
    if (nMP > 0) HIP_TRY(hipMemcpyAsync(o->dX[0], o->dBa + oX, (size_t)nMP * 24, hipMemcpyDeviceToDevice, st));
    if (nE > 0) HIP_TRY(hipMemsetAsync(o->dEOff, 0, (size_t)nE, st));
    const double tC = now();
    // part B, pass 2: prefix sums, then every edge into its landmark's list and its key-frame's rows (stable: input order inside a group)
    for (int p2 = 0; p2 < nMP; p2++) ptStart[p2 + 1] += ptStart[p2];
    for (int c = 0; c < nOpt; c++) kfRowStart[c + 1] += kfRowStart[c];
    o->hFill.resize((size_t)nMP + nOpt + 2);
    {
        int32_t *fillP = o->hFill.data(), *fillK = fillP + nMP + 1;
        std::memcpy(fillP, ptStart, (size_t)nMP * 4);
        std::memcpy(fillK, kfRowStart, (size_t)nOpt * 4);
        for (int e = 0; e < nE; e++) {
            ptEdge[fillP[e_mp[e]]++] = e;
            const int c = poseCol[e_kf[e]];
            int slot = -1;
            if (c >= 0) { slot = fillK[c]; fillK[c] = slot + 2; }
            rowSlot[e] = slot;
        }
    }
    HIP_TRY(hipMemcpyAsync(o->dBa + partA, hs + partA, upBytes - partA, hipMemcpyHostToDevice, st));
    const double tD = now();
    BADev B{};
    B.nKF = nKF; B.nMP = nMP; B.nE = nE; B.nOpt = nOpt; B.n = n;
    B.eMP = (const int32_t *)(o->dBa + oEM); B.eKF = (const int32_t *)(o->dBa + oEK); B.poseCol = (const int32_t *)(o->dBa + oPC);
    B.ptStart = (const int32_t *)(o->dBa + oPS); B.ptEdge = (const int32_t *)(o->dBa + oPE); B.rowSlot = (const int32_t *)(o->dBa + oRS);
    B.kfRowStart = (const int32_t *)(o->dBa + oKR); B.obs = (const float *)(o->dBa + oOb); B.info = (const float *)(o->dBa + oIn);
    B.cam = DCam{K4[0], K4[1], K4[2], K4[3]};
    B.delta = mode == 0 ? (double)(float)std::sqrt(5.991) : (double)(float)std::sqrt(5.99);   // thHuberMono = sqrt(5.991) / thHuber2D = sqrt(5.99)
    B.dsqr = B.delta * B.delta;
    B.off = o->dEOff; B.robust = mode == 2 ? (gbaRobust ? 1 : 0) : 1;
    B.Hll = o->dHll; B.bl = o->dBl; B.Hpl = o->dHpl; B.panel = o->dPanel; B.Hpp = o->dHpp; B.bp = o->dBp; B.Dinv = o->dDinv; B.S = o->dS;
    B.bs = o->dBs; B.x = o->dXv; B.lastChi2 = o->dChi; B.scal = o->dScal;

    const int gE = std::max(1, (nE + 255) / 256);
    // up to 42 optimised key-frames (255 unknowns): dense Schur panel on the matrix cores + one-workgroup solve; beyond: block-sparse Schur
    // accumulation + multi-workgroup blocked Cholesky (the "big" kernels above)
    const bool big = n > 255;
    const int NP = big ? 16 : (n + 1 + 15) / 16 * 16, NT = NP / 16, K3 = 3 * nMP;
    if (!big && NP > o->npCap) { g_lastError = "local BA: reduced system larger than the optimiser's arenas"; return RUMI_E_CAPACITY; }
    int nBlocks = 0;
    if (big) {
        if ((size_t)n * 8 > 120 * 1024) { g_lastError = "bundle adjustment: more than 2560 optimised key-frames"; return RUMI_E_CAPACITY; }
        // observation pairs of every landmark grouped by Schur block (ca, cb <= ca): counting sort over the blocks
        std::vector<int32_t> colAt((size_t)std::max(nE, 1));             // column block of the t-th entry of ptEdge (-1 fixed)
        for (int t = 0; t < nE; t++) colAt[t] = poseCol[e_kf[ptEdge[t]]];
        std::vector<int64_t> cnt((size_t)nOpt * nOpt + 1, 0);
        auto for_pairs = [&](auto &&f) {
            for (int p = 0; p < nMP; p++) {
                const int t0 = ptStart[p], t1 = ptStart[p + 1];
                for (int ia = t0; ia < t1; ia++) {
                    const int ca = colAt[ia];
                    if (ca < 0) continue;
                    const size_t rowKey = (size_t)ca * nOpt;
                    for (int ib = t0; ib < t1; ib++) {
                        const int cb = colAt[ib];
                        if (cb < 0 || cb > ca || (cb == ca && ib != ia)) continue;
                        f(rowKey + cb, ia, ib);
                    }
                }
            }
        };
        for_pairs([&](size_t key, int, int) { cnt[key + 1]++; });
        std::vector<int32_t> blk;
        for (size_t key = 0; key < (size_t)nOpt * nOpt; key++) {
            const int64_t c0 = cnt[key], c1 = cnt[key] + cnt[key + 1];
            for (int64_t a0 = c0; a0 < c1; a0 += kSchurSeg) {
                blk.push_back((int32_t)(key / nOpt)); blk.push_back((int32_t)(key % nOpt) | (c1 - c0 > kSchurSeg ? 1 << 30 : 0));
                blk.push_back((int32_t)a0); blk.push_back((int32_t)std::min<int64_t>(a0 + kSchurSeg, c1));
            }
            cnt[key + 1] += cnt[key];
        }
        const int64_t nPairs = cnt[(size_t)nOpt * nOpt];
        if (nPairs > (int64_t)1 << 30) { g_lastError = "bundle adjustment: more than 2^30 co-observation pairs"; return RUMI_E_CAPACITY; }
        std::vector<int32_t> pairs((size_t)std::max<int64_t>(nPairs, 1) * 2);
        for_pairs([&](size_t key, int ia, int ib) { const int64_t at = cnt[key]++; pairs[2 * at] = ptEdge[ia]; pairs[2 * at + 1] = ptEdge[ib]; });
        nBlocks = (int)(blk.size() / 4);
        const size_t need = (blk.size() + pairs.size()) * sizeof(int32_t);
        if (need > o->pairCap) {
            if (o->dPairs) (void)hipFree(o->dPairs);
            o->dPairs = nullptr; o->pairCap = 0;
            HIP_TRY(hipMalloc((void **)&o->dPairs, need + need / 4));
            o->pairCap = need + need / 4;
        }
        o->pairOff = blk.size();
        if (!blk.empty()) HIP_TRY(hipMemcpyAsync(o->dPairs, blk.data(), blk.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(o->dPairs + blk.size(), pairs.data(), pairs.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));                 // blk / pairs are temporaries
        if (!o->dW) { int rcw = oalloc(&o->dW, (size_t)o->maxE * 18); if (rcw == RUMI_OK) rcw = oalloc(&o->dColOf, (size_t)o->maxE); if (rcw != RUMI_OK) return rcw; }
        if ((size_t)n * 8 > 16 * 1024)
            HIP_TRY(raise_lds_limit(reinterpret_cast<const void *>(k_chol_backsub), (size_t)n * 8));
    }
    const size_t ldsSolve = ((size_t)(n + 1) * (n + 1) + (size_t)n) * sizeof(double);
    const int useLds = !big && ldsSolve <= 158 * 1024;
    // tile formulation (k_ba_solve_tiles) whenever its tiles fit in LDS; RUMI_BA_SOLVE_PANEL8=1 keeps the panel-8 kernel (A/B measurements)
    static const bool forcePanel8 = std::getenv("RUMI_BA_SOLVE_PANEL8") != nullptr;
    const size_t ldsTiles = ((size_t)(NT * (NT + 1) / 2) * 256 + (size_t)NT * 16) * sizeof(double);
    const int useTiles = !big && n > 0 && NT <= kSolveTilesMax && !forcePanel8;
    if (useTiles && ldsTiles > 48 * 1024)
        HIP_TRY(raise_lds_limit(reinterpret_cast<const void *>(k_ba_solve_tiles), ldsTiles));
    // more than 64 KiB of dynamic LDS needs the opt-in; the limit is process state and only grows (rumi_common.h: raise_lds_limit), so that the
    // worker threads of rumi_local_ba_batch and the facade's per-thread arenas cannot lower it under each other
    if (useLds && ldsSolve > 48 * 1024)
        HIP_TRY(raise_lds_limit(reinterpret_cast<const void *>(k_ba_solve<true>), ldsSolve));
    const int nSlices = 64;
    // the eight scalars of o->dScal -> o->hScal, without a runtime synchronisation (see k_ba_publish); falls back to one if the stream has
    // drained without the number arriving (a failed launch)
    auto fetch_scalars = [&]() -> int {
        const unsigned long long seq = ++o->pubSeq;
        hipLaunchKernelGGL(k_ba_publish, dim3(1), dim3(64), 0, st, o->dScal, o->dhScal, seq);
        HIP_TRY(hipGetLastError());
        volatile unsigned long long *flag = reinterpret_cast<volatile unsigned long long *>(o->hScal + 8);
        for (unsigned spin = 0; *flag != seq; spin++) {
            if ((spin & 0xFFFF) == 0xFFFF && hipStreamQuery(st) != hipErrorNotReady) {
                HIP_TRY(hipStreamSynchronize(st));
                if (*flag != seq) { HIP_TRY(hipMemcpyAsync(o->hScal, o->dScal, 8 * sizeof(double), hipMemcpyDeviceToHost, st)); HIP_TRY(hipStreamSynchronize(st)); break; }
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        return RUMI_OK;
    };
    auto chi2_of = [&](int which, double *out) -> int {
        HIP_TRY(hipMemsetAsync(o->dScal, 0, sizeof(double), st));
        hipLaunchKernelGGL(k_ba_chi2, dim3(gE), dim3(256), 0, st, B, o->dT[which], o->dX[which]);
        { const int rcf = fetch_scalars(); if (rcf != RUMI_OK) return rcf; }
        *out = o->hScal[0];
        return RUMI_OK;
    };
    if (nMP > 0 && !big) HIP_TRY(hipMemsetAsync(o->dYt, 0, (size_t)K3 * NP * sizeof(double), st));   // pattern of Y is fixed: zero once, live entries are rewritten per trial
    // reduced system of one LM trial -> B.x, B.scal[3]
    auto solve_big = [&](double lambda) -> int {
        double *A = o->dAglob, *rdg = A + (size_t)(n + 1) * n;
        const int ld = n;
        const size_t tot = (size_t)(n + 1) * n;
        hipLaunchKernelGGL(k_big_init, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, B, lambda, A, ld);
        if (nMP > 0) {
            BADev Bz = B;
            Bz.n = 0;                                                      // z = L^T b_l lands in dYt[3 p .. 3 p + 2]
            hipLaunchKernelGGL(k_ba_dinv, dim3((nMP + 255) / 256), dim3(256), 0, st, Bz, lambda, o->dYt, 1, o->dLp);
            if (nE > 0) {
                hipLaunchKernelGGL(k_big_w, dim3(gE), dim3(256), 0, st, B, o->dLp, o->dW, o->dColOf);
                if (nBlocks > 0) hipLaunchKernelGGL(k_big_schur, dim3((nBlocks + 3) / 4), dim3(256), 0, st, B, o->dPairs, nBlocks, o->dPairs + o->pairOff, o->dW, o->dYt, A, ld);
            }
        }
        for (int j0 = 0; j0 < n; j0 += kNB) {
            const int w = std::min(kNB, n - j0), rows = n + 1 - (j0 + w);
            hipLaunchKernelGGL(k_chol_diag, dim3(1), dim3(256), 0, st, A, ld, n, j0, rdg, o->dScal);
            if (rows > 0) {
                hipLaunchKernelGGL(k_chol_trsm, dim3((rows + 63) / 64), dim3(256), 0, st, A, ld, n, j0, rdg);
                const int T = (rows + 63) / 64;
                if (j0 + w < n) hipLaunchKernelGGL(k_chol_syrk, dim3(T, T), dim3(256), 0, st, A, ld, n, j0);
            }
        }
        hipLaunchKernelGGL(k_chol_backsub, dim3(1), dim3(1024), (size_t)n * sizeof(double), st, A, ld, n, rdg, o->dXv, o->dScal);
        HIP_TRY(hipGetLastError());
        return RUMI_OK;
    };
    HIP_TRY(hipEventRecord(o->ev[0], st));
    const bool prof = o->profiling;
    bool hppFresh = false, gClean = false;
    for (auto &k : o->kernelMs) k = 0.f;
    int cur = 0, iters = 0, trials = 0, rc = RUMI_OK;
    bool ranChi2 = false;
    auto lm = [&](int maxIt) -> int {                       // g2o optimize(maxIt)
    double lambda = -1, ni = 2;
    int nBad = 0;
    double currentChi = 0;
    for (int it = 0; it < maxIt && !(stop_flag && *stop_flag); it++) {
        // g2o recomputes the active errors here; the value is already known after the first iteration (an accepted trial
        // left it in tempChi, a rejected one did not change the state), so only the first iteration launches the kernel.
        if (it == 0 && (rc = chi2_of(cur, &currentChi)) != RUMI_OK) return rc;
        ranChi2 = true;
        const double iniChi = currentChi;
        // buildSystem
        {
            const ZeroList Z{{o->dHll, o->dBl, o->dHpp, o->dBp}, {nMP * 9, nMP * 3, nOpt * 36, n}};
            const int zmax = std::max(std::max(nMP * 9, nOpt * 36), 1);
            hipLaunchKernelGGL(k_ba_zero, dim3((zmax + 255) / 256), dim3(256), 0, st, Z);
        }
        if (nE > 0) hipLaunchKernelGGL(k_ba_build, dim3(gE), dim3(256), 0, st, B, o->dT[cur], o->dX[cur]);
        if (prof) HIP_TRY(hipEventRecord(o->evK[0], st));
        if (nOpt > 0) hipLaunchKernelGGL(k_ba_hpp_mfma, dim3(nOpt, kHppSlices), dim3(256), 0, st, B);
        if (prof) { HIP_TRY(hipEventRecord(o->evK[1], st)); hppFresh = true; }
        if (it == 0) {
            HIP_TRY(hipMemsetAsync(o->dScal + 2, 0, sizeof(double), st));
            const int nd = nOpt * 6 + nMP * 3;
            hipLaunchKernelGGL(k_ba_maxdiag, dim3((nd + 255) / 256), dim3(256), 0, st, B);
            if ((rc = fetch_scalars()) != RUMI_OK) return rc;
            lambda = 1e-5 * o->hScal[2]; ni = 2; nBad = 0;
        }
        double rho = 0;
        int qmax = 0;
        do {
            const int trial = cur ^ 1;
            if (!useTiles || !gClean) {                        // k_ba_solve_tiles leaves G and the two accumulators of dScal zeroed itself
                const ZeroList Z{{o->dG, o->dScal, nullptr, nullptr}, {big ? 0 : NP * NP, 2, 0, 0}};
                hipLaunchKernelGGL(k_ba_zero, dim3(((big ? 2 : NP * NP) + 255) / 256), dim3(256), 0, st, Z);
                gClean = true;
            }
            if (big) { if ((rc = solve_big(lambda)) != RUMI_OK) return rc; }
            else {
            if (nMP > 0) hipLaunchKernelGGL(k_ba_dinv_yfill, dim3((nMP + nE + 255) / 256), dim3(256), 0, st, B, lambda, o->dYt, NP, o->dLp);
            if (prof) HIP_TRY(hipEventRecord(o->evK[2], st));
            if (nMP > 0 && n > 0) hipLaunchKernelGGL(k_ba_syrk_mfma, dim3(NT * (NT + 1) / 2 * (nSlices / 4)), dim3(256), 0, st, o->dYt, K3, NP, nSlices, o->dG);
            if (prof) { HIP_TRY(hipEventRecord(o->evK[3], st)); HIP_TRY(hipEventRecord(o->evK[4], st)); }
            if (n > 0) {
                if (useTiles) hipLaunchKernelGGL(k_ba_solve_tiles, dim3(1), dim3(kSolveThreads), ldsTiles, st, B, lambda, o->dG, NP);
                else if (useLds) hipLaunchKernelGGL(k_ba_solve<true>, dim3(1), dim3(1024), ldsSolve, st, B, lambda, o->dG, NP, o->dAglob);
                else hipLaunchKernelGGL(k_ba_solve<false>, dim3(1), dim3(1024), 0, st, B, lambda, o->dG, NP, o->dAglob);
            }
            else HIP_TRY(hipMemsetAsync(o->dScal + 3, 0, sizeof(double), st));
            if (prof) HIP_TRY(hipEventRecord(o->evK[5], st));
            }
            hipLaunchKernelGGL(k_ba_update, dim3(((nMP + nKF) * kLmLanes + 255) / 256), dim3(256), 0, st, B, lambda, o->dT[cur], o->dX[cur], o->dT[trial], o->dX[trial]);
            hipLaunchKernelGGL(k_ba_chi2, dim3(gE), dim3(256), 0, st, B, o->dT[trial], o->dX[trial]);
            HIP_TRY(hipGetLastError());
            if (prof) HIP_TRY(hipStreamSynchronize(st));              // the event pairs below must have completed
            if ((rc = fetch_scalars()) != RUMI_OK) return rc;
            if (prof && !big) {
                float ms;
                if (hppFresh) { HIP_TRY(hipEventElapsedTime(&ms, o->evK[0], o->evK[1])); o->kernelMs[0] += ms; hppFresh = false; }
                HIP_TRY(hipEventElapsedTime(&ms, o->evK[2], o->evK[3])); o->kernelMs[1] += ms;
                HIP_TRY(hipEventElapsedTime(&ms, o->evK[4], o->evK[5])); o->kernelMs[2] += ms;
                o->kernelMs[3] += 1.f;
            }
            const bool ok2 = n == 0 || o->hScal[3] != 0.0;
            double tempChi = o->hScal[0];
            if (!ok2) tempChi = std::numeric_limits<double>::max();
            rho = currentChi - tempChi;
            const double scale = o->hScal[1] + 1e-3;
            rho /= scale;
            if (rho > 0 && std::isfinite(tempChi)) {
                double alpha = 1. - std::pow((2 * rho - 1), 3);
                alpha = std::min(alpha, 2. / 3.);
                lambda *= std::max(1. / 3., alpha);
                ni = 2;
                currentChi = tempChi;
                cur = trial;                       // discardTop(): the trial state becomes the estimate
            } else {
                lambda *= ni;
                ni *= 2;                           // pop(): keep the current state
            }
            qmax++;
            trials++;
        } while (rho < 0 && qmax < 10 && !(stop_flag && *stop_flag));
        iters++;
        if (qmax == 10 || rho == 0) break;
        if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
        if (nBad >= 3) break;
    }
    return RUMI_OK;
    };
    // The same loop with the decisions taken on the device (LmState / k_lm_decide above): slots of one LM trial each are enqueued one ahead of the
    // device, no kernel waits for the host.  For the windows the tile solver handles (up to 29 optimised key-frames); the large problems of
    // global BA, the panel solvers and the profiled path keep the host loop.  RUMI_BA_HOST_LM=1 forces the host loop (A/B measurements).
    static const bool forceHostLm = std::getenv("RUMI_BA_HOST_LM") != nullptr;
    const bool deviceLm = !big && useTiles && !prof && !forceHostLm && nE > 0;
    auto lm_device = [&](int maxIt) -> int {
        if (stop_flag && *stop_flag) return RUMI_OK;                       // the host loop's condition before its first iteration
        LmState &init = *o->hLmInit;                                       // (pinned; the previous run's copy has long completed)
        init = LmState{};
        init.T[0] = o->dT[0]; init.T[1] = o->dT[1]; init.X[0] = o->dX[0]; init.X[1] = o->dX[1];
        init.cur = cur; init.needBuild = 1; init.maxIt = maxIt; init.nUnknowns = n; init.ni = 2; init.lambda = -1;
        *o->hStop = 0;
        // slots of an earlier run may still be queued (they exit at once, but their k_lm_decide still reports): only reports carrying a sequence
        // number of THIS run count
        const unsigned long long firstSeq = o->lmSeq + 1;
        auto report = [&](int32_t *trialsDone, int32_t *done) {
            const unsigned long long q = __atomic_load_n(const_cast<const unsigned long long *>(&o->hLm->seq), __ATOMIC_ACQUIRE);
            if (q < firstSeq) { *trialsDone = 0; *done = 0; return; }
            *trialsDone = o->hLm->trialsDone; *done = o->hLm->done;
        };
        HIP_TRY(hipMemcpyAsync(o->dLm, &init, sizeof init, hipMemcpyHostToDevice, st));
        const LmState *S = o->dLm;
        // prologue of the first iteration: chi2 of the estimate, clean accumulators, buildSystem, computeLambdaInit
        {
            const ZeroList Z0{{o->dG, o->dScal, nullptr, nullptr}, {gClean ? 0 : NP * NP, 3, 0, 0}};
            hipLaunchKernelGGL(k_ba_zero, dim3(((gClean ? 3 : NP * NP) + 255) / 256), dim3(256), 0, st, Z0, (const LmState *)nullptr);
            gClean = true;
            hipLaunchKernelGGL(k_ba_chi2, dim3(gE), dim3(256), 0, st, B, o->dT[cur], o->dX[cur], S, 0);
            ranChi2 = true;
        }
        const ZeroList Zb{{o->dHll, o->dBl, o->dHpp, o->dBp}, {nMP * 9, nMP * 3, nOpt * 36, n}};
        const int zmax = std::max(std::max(nMP * 9, nOpt * 36), 1);
        auto enqueue_slot = [&](bool first) {
            hipLaunchKernelGGL(k_ba_zero, dim3((zmax + 255) / 256), dim3(256), 0, st, Zb, S);
            hipLaunchKernelGGL(k_ba_build, dim3(gE), dim3(256), 0, st, B, o->dT[0], o->dX[0], S);
            if (nOpt > 0) hipLaunchKernelGGL(k_ba_hpp_mfma, dim3(nOpt, kHppSlices), dim3(256), 0, st, B, S);
            if (first) {
                const int nd = nOpt * 6 + nMP * 3;
                hipLaunchKernelGGL(k_ba_maxdiag, dim3((nd + 255) / 256), dim3(256), 0, st, B);
                hipLaunchKernelGGL(k_lm_begin, dim3(1), dim3(64), 0, st, o->dScal, o->dLm);
            }
            if (nMP > 0) hipLaunchKernelGGL(k_ba_dinv_yfill, dim3((nMP + nE + 255) / 256), dim3(256), 0, st, B, 0.0, o->dYt, NP, o->dLp, S);
            if (nMP > 0 && n > 0) hipLaunchKernelGGL(k_ba_syrk_mfma, dim3(NT * (NT + 1) / 2 * (nSlices / 4)), dim3(256), 0, st, o->dYt, K3, NP, nSlices, o->dG, S);
            hipLaunchKernelGGL(k_ba_solve_tiles, dim3(1), dim3(kSolveThreads), ldsTiles, st, B, 0.0, o->dG, NP, S);
            hipLaunchKernelGGL(k_ba_update, dim3(((nMP + nKF) * kLmLanes + 255) / 256), dim3(256), 0, st, B, 0.0, o->dT[0], o->dX[0], o->dT[1], o->dX[1], S);
            hipLaunchKernelGGL(k_ba_chi2, dim3(gE), dim3(256), 0, st, B, o->dT[0], o->dX[0], S, 1);
            hipLaunchKernelGGL(k_lm_decide, dim3(1), dim3(64), 0, st, o->dScal, o->dLm, o->dhLm, o->dhStop, ++o->lmSeq);
        };
        // one slot queued ahead of the one that is running.  While it waits the host thread SLEEPS in short naps (a slot lasts ~100 us; the
        // thread's timer slack is 1 us meanwhile, so that a 30 us nap is not rounded up to the default 50 us slack): the workers of
        // rumi_local_ba_batch no longer burn a core each (hipEventSynchronize spins on this runtime whatever the event's flags; measured).
        // The reference's stop flag is forwarded once per nap.
        // (the calling thread's own slack is put back when the loop is over: it is the caller's thread)
        struct SlackGuard {
            long old;
            SlackGuard() : old(prctl(PR_GET_TIMERSLACK, 0UL, 0UL, 0UL, 0UL)) { (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0UL, 0UL, 0UL); }
            ~SlackGuard() { if (old > 0) (void)prctl(PR_SET_TIMERSLACK, (unsigned long)old, 0UL, 0UL, 0UL); }
        } slackGuard;
        const int maxSlots = maxIt * 10;
        int enq = 0, naps = 0;
        for (;;) {
            int32_t tdone = 0, done = 0;
            report(&tdone, &done);
            if (done) break;
            if (stop_flag && *stop_flag) *o->hStop = 1;
            if (enq < maxSlots && enq <= tdone + 1) {
                enqueue_slot(enq == 0);
                HIP_TRY(hipGetLastError());
                enq++;
                naps = 0;
                continue;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(30));
            if ((++naps & 0xFF) == 0 && hipStreamQuery(st) != hipErrorNotReady) {
                // the stream has drained: either the last report is about to be seen, or a launch failed
                HIP_TRY(hipStreamSynchronize(st));
                report(&tdone, &done);
                if (!done && (enq >= maxSlots || enq > tdone + 1)) { g_lastError = "local BA: the device-side LM loop stalled"; return RUMI_E_NO_DEVICE; }
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        cur = o->hLm->cur; iters += o->hLm->iters; trials += o->hLm->trials;
        return RUMI_OK;
    };
    auto lm_any = [&](int maxIt) -> int { return deviceLm ? lm_device(maxIt) : lm(maxIt); };
    int itersFirst = 0;
    const double tE = now();
    if ((rc = lm_any(mode == 0 ? 10 : mode == 1 ? 5 : gbaIterations)) != RUMI_OK) return rc;
    const double tF = now();
    itersFirst = iters;
    if (mode == 1 && !(stop_flag && *stop_flag)) {          // bDoMore
        if (nE > 0 && ranChi2) hipLaunchKernelGGL(k_ba_mark, dim3(gE), dim3(256), 0, st, B, o->dT[cur], o->dX[cur], o->dEOff);
        B.robust = 0;
        if ((rc = lm_any(10)) != RUMI_OK) return rc;
    }
    // results gathered by the last kernel into one block and read back with one copy: [T | X | erase].  (Taking a second stream for this, past the
    // slot the device-side LM loop has enqueued ahead -- nine launches that return at once -- measured no faster for one window and 7 % slower
    // for batches of windows: by the time the host has seen the loop end, that slot has run.)
    const size_t rT = 0, rX = al(rT + (size_t)nKF * 64), rE = al(rX + (size_t)nMP * 24), dnBytes = al(rE + (size_t)nE);
    hipStream_t se = st;
    const int gF = std::max(gE, std::max((nKF * 8 + 255) / 256, (nMP * 3 + 255) / 256));
    hipLaunchKernelGGL(k_ba_finalize, dim3(gF), dim3(256), 0, se, B, o->dT[cur], o->dX[cur], ranChi2 ? 1 : 0, o->dBaOut + rE, reinterpret_cast<double *>(o->dBaOut + rT),
                       reinterpret_cast<double *>(o->dBaOut + rX));
    HIP_TRY(hipEventRecord(o->ev[1], se));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(o->hBa, o->dBaOut, dnBytes, hipMemcpyDeviceToHost, se));
    HIP_TRY(hipStreamSynchronize(se));
    const double *T1 = reinterpret_cast<const double *>(o->hBa + rT), *X1 = reinterpret_cast<const double *>(o->hBa + rX);
    if (nE > 0) std::memcpy(erase_out, o->hBa + rE, (size_t)nE);
    HIP_TRY(hipEventElapsedTime(&o->stageMs[5], o->ev[0], o->ev[1]));
    for (int k = 0; k < nKF; k++) {
        if (kf_fixed[k]) continue;
        const double *t = T1 + (size_t)k * 8;
        se3_to_float7(DSE3{{t[0], t[1], t[2], t[3]}, {t[4], t[5], t[6]}}, kf_pose7 + (size_t)k * 7);
    }
    for (size_t i = 0; i < (size_t)nMP * 3; i++) mp_pos3[i] = (float)X1[i];
    if (hostDbg) fprintf(stderr, "ba host us: pass1+partA %.1f h2d-A %.1f pass2+h2d-B %.1f setup %.1f lm %.1f tail %.1f\n", tB - tA, tC - tB, tD - tC, tE - tD, tF - tE, now() - tF);
    if (stats) { stats[0] = mode == 1 ? itersFirst : iters; stats[1] = trials; stats[2] = nOpt; stats[3] = mode == 1 ? iters - itersFirst : 0; }
    return RUMI_OK;
}

extern "C" int rumi_local_ba(RumiOptimizer *o, int32_t nKF, float *kf_pose7, const uint8_t *kf_fixed, int32_t nMP, float *mp_pos3,
                             int32_t nE, const int32_t *e_mp, const int32_t *e_kf, const float *e_obs, const float *e_inv_sigma2,
                             const float *K4, const volatile uint8_t *stop_flag, uint8_t *erase_out, int32_t *stats) {
    return ba_run(o, 0, nKF, kf_pose7, kf_fixed, nMP, mp_pos3, nE, e_mp, e_kf, e_obs, e_inv_sigma2, K4, stop_flag, erase_out, stats);
}

// R independent local windows (the only multi-window form local BA has: a window does not shard).  n_workers host threads, each
// with a child handle of its own (own stream, own arenas, created on first use and kept), take the windows from a shared counter: while one
// window's host thread waits for the eight scalars of an LM trial, the kernels of the others fill the device.
extern "C" int rumi_local_ba_batch(RumiOptimizer *o, int32_t n_windows, RumiBaWindow *win, int32_t n_workers) {
    if (!o || n_windows < 0 || (n_windows > 0 && !win) || n_workers < 1) return RUMI_E_INVALID;
    if (n_windows == 0) return RUMI_OK;
    auto need_children = [&](int cnt) -> int {
        while ((int)o->workers.size() < cnt) {
            RumiOptimizer *c = nullptr;
            const int rc = rumi_opt_create(o->maxPoseEdges, 1, o->maxKF, o->maxMP, o->maxE, o->device, &c);
            if (rc != RUMI_OK) return rc;
            o->workers.push_back(c);
        }
        return RUMI_OK;
    };
    // windows of up to 29 optimised key-frames: the window is a batch dimension of the kernels (ba_windows.inc), driven by THIS thread alone, in
    // groups of kBawMaxWindows; a child handle per window of a group lends its arenas
    std::vector<int> batched, others;
    for (int i = 0; i < n_windows; i++) {
        const RumiBaWindow &W = win[i];
        const bool ok = W.kf_pose7 && W.kf_fixed && W.K4 && W.mp_pos3 && W.e_mp && W.e_kf && W.e_obs && W.e_inv_sigma2 && W.erase_out &&
                        W.n_kf <= o->maxKF && W.n_mp <= o->maxMP && W.n_edges <= o->maxE && !o->profiling && !std::getenv("RUMI_BA_HOST_LM") &&
                        baw_eligible(W.n_kf, W.kf_fixed, W.n_mp, W.n_edges);
        (ok ? batched : others).push_back(i);
    }
    for (size_t g0 = 0; g0 < batched.size(); g0 += kBawMaxWindows) {
        const int cnt = (int)std::min<size_t>(kBawMaxWindows, batched.size() - g0);
        { const int rc = need_children(cnt); if (rc != RUMI_OK) return rc; }
        std::vector<BawArgs> args((size_t)cnt);
        std::vector<int32_t> status((size_t)cnt, RUMI_OK);
        for (int j = 0; j < cnt; j++) {
            RumiBaWindow &W = win[batched[g0 + j]];
            args[j] = BawArgs{W.n_kf, W.kf_pose7, W.kf_fixed, W.n_mp, W.mp_pos3, W.n_edges, W.e_mp, W.e_kf, W.e_obs, W.e_inv_sigma2, W.K4, W.stop_flag, W.erase_out, W.stats};
        }
        // launch groups (two from 8 windows on, three from 12: baw_run): the parent handle lends the first its stream and window table, further children the others
        { const int rc = need_children(cnt + 3); if (rc != RUMI_OK) return rc; }
        RumiOptimizer *runners[4] = {o, o->workers[cnt], o->workers[cnt + 1], o->workers[cnt + 2]};
        const int rc = baw_run(o, 0, cnt, args.data(), o->workers.data(), 0, 1, status.data(), runners, 4);
        for (int j = 0; j < cnt; j++) win[batched[g0 + j]].status = status[j];
        if (rc != RUMI_OK && rc != RUMI_E_INVALID && rc != RUMI_E_CAPACITY) return rc;      // a HIP failure: nothing more to run
    }
    // everything else (larger windows, structure-only windows, the profiled path): one single-window run each, over worker threads as before
    if (!others.empty()) {
        n_workers = std::min(std::min(n_workers, (int)others.size()), 16);
        { const int rc = need_children(n_workers); if (rc != RUMI_OK) return rc; }
        std::atomic<int> next{0};
        auto work = [&](RumiOptimizer *c) {
            for (int q = next.fetch_add(1); q < (int)others.size(); q = next.fetch_add(1)) {
                RumiBaWindow &W = win[others[q]];
                W.status = ba_run(c, 0, W.n_kf, W.kf_pose7, W.kf_fixed, W.n_mp, W.mp_pos3, W.n_edges, W.e_mp, W.e_kf, W.e_obs, W.e_inv_sigma2, W.K4, W.stop_flag,
                                  W.erase_out, W.stats);
            }
        };
        std::vector<std::thread> th;
        for (int k = 1; k < n_workers; k++) th.emplace_back(work, o->workers[k]);
        work(o->workers[0]);
        for (auto &t : th) t.join();
    }
    int worst = RUMI_OK;
    for (int i = 0; i < n_windows; i++) if (win[i].status != RUMI_OK) worst = win[i].status;
    return worst;
}

extern "C" int rumi_bundle_adjustment(RumiOptimizer *o, int32_t nKF, float *kf_pose7, const uint8_t *kf_fixed, int32_t nMP, float *mp_pos3,
                                      int32_t nE, const int32_t *e_mp, const int32_t *e_kf, const float *e_obs, const float *e_inv_sigma2,
                                      const float *K4, const volatile uint8_t *stop_flag, int32_t n_iterations, int32_t robust, int32_t *stats) {
    if (n_iterations < 0) return RUMI_E_INVALID;
    std::vector<uint8_t> erase((size_t)std::max(nE, 1));
    return ba_run(o, 2, nKF, kf_pose7, kf_fixed, nMP, mp_pos3, nE, e_mp, e_kf, e_obs, e_inv_sigma2, K4, stop_flag, erase.data(), stats, n_iterations, robust);
}

extern "C" int rumi_merge_ba(RumiOptimizer *o, int32_t nKF, float *kf_pose7, const uint8_t *kf_fixed, int32_t nMP, float *mp_pos3,
                             int32_t nE, const int32_t *e_mp, const int32_t *e_kf, const float *e_obs, const float *e_inv_sigma2,
                             const float *K4, const volatile uint8_t *stop_flag, uint8_t *erase_out, int32_t *stats) {
    return ba_run(o, 1, nKF, kf_pose7, kf_fixed, nMP, mp_pos3, nE, e_mp, e_kf, e_obs, e_inv_sigma2, K4, stop_flag, erase_out, stats);
}


extern "C" int rumi_sim3_inliers(RumiOptimizer *o, int32_t n_pairs, const int32_t *pair_start, const int32_t *pair_denominator,
                                 const double *S_c1w2, const double *S_c2w1, const float *K4_1, const float *K4_2, const float *X1, const float *X2,
                                 const float *kp1, const float *kp2, const float *sigma2_1, const float *sigma2_2, const uint8_t *edge1,
                                 const uint8_t *edge2, uint8_t *inlier_out, float *ratio_out, float *median_out) {
    if (!o || n_pairs < 0 || !pair_start || !pair_denominator || !K4_1 || !K4_2 || !median_out) return RUMI_E_INVALID;
    *median_out = 0.f;
    if (n_pairs == 0) return RUMI_OK;
    const int total = pair_start[n_pairs];
    if (total < 0 || (total > 0 && (!S_c1w2 || !S_c2w1 || !X1 || !X2 || !kp1 || !kp2 || !sigma2_1 || !sigma2_2 || !edge1 || !edge2 || !inlier_out)))
        return RUMI_E_INVALID;
    std::vector<float> ratio(n_pairs, 0.f);
    if (total > 0) {
        HIP_TRY(hipSetDevice(o->device));
        // one pinned block up (read in place), one flag array back
        auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
        const size_t oP = 0, oA = al(oP + (size_t)total * 4), oB = al(oA + (size_t)n_pairs * 64), oK = al(oB + (size_t)n_pairs * 64), oX1 = al(oK + 32),
                     oX2 = al(oX1 + (size_t)total * 12), oK1 = al(oX2 + (size_t)total * 12), oK2 = al(oK1 + (size_t)total * 8), oS1 = al(oK2 + (size_t)total * 8),
                     oS2 = al(oS1 + (size_t)total * 4), oE1 = al(oS2 + (size_t)total * 4), oE2 = al(oE1 + (size_t)total), bytes = al(oE2 + (size_t)total);
        if (bytes > o->baStageCap || (size_t)total > o->baStageCap) { g_lastError = "Sim3 inliers: more matches than the optimiser's arenas hold"; return RUMI_E_CAPACITY; }
        uint8_t *h = o->hBa;
        int32_t *pairOf = reinterpret_cast<int32_t *>(h + oP);
        for (int p = 0; p < n_pairs; p++) for (int i = pair_start[p]; i < pair_start[p + 1]; i++) pairOf[i] = p;
        std::memcpy(h + oA, S_c1w2, (size_t)n_pairs * 64); std::memcpy(h + oB, S_c2w1, (size_t)n_pairs * 64);
        std::memcpy(h + oK, K4_1, 16); std::memcpy(h + oK + 16, K4_2, 16);
        std::memcpy(h + oX1, X1, (size_t)total * 12); std::memcpy(h + oX2, X2, (size_t)total * 12);
        std::memcpy(h + oK1, kp1, (size_t)total * 8); std::memcpy(h + oK2, kp2, (size_t)total * 8);
        std::memcpy(h + oS1, sigma2_1, (size_t)total * 4); std::memcpy(h + oS2, sigma2_2, (size_t)total * 4);
        std::memcpy(h + oE1, edge1, (size_t)total); std::memcpy(h + oE2, edge2, (size_t)total);
        HIP_TRY(hipMemcpyAsync(o->dBa, h, bytes, hipMemcpyHostToDevice, nullptr));
        uint8_t *d = o->dBa;
        hipLaunchKernelGGL(k_sim3_inliers, dim3((total + 255) / 256), dim3(256), 0, nullptr, total, (const int32_t *)(d + oP), (const double *)(d + oA),
                           (const double *)(d + oB), (const float *)(d + oK), (const float *)(d + oK + 16), (const float *)(d + oX1), (const float *)(d + oX2),
                           (const float *)(d + oK1), (const float *)(d + oK2), (const float *)(d + oS1), (const float *)(d + oS2), d + oE1, d + oE2, o->dBaOut);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(inlier_out, o->dBaOut, (size_t)total, hipMemcpyDeviceToHost));
    }
    for (int p = 0; p < n_pairs; p++) {                                       // :643-646
        int nIn = 0;
        for (int i = pair_start[p]; i < pair_start[p + 1]; i++) nIn += inlier_out[i];
        ratio[p] = pair_denominator[p] ? (float)nIn / (float)pair_denominator[p] : 0.f;
    }
    if (ratio_out) std::memcpy(ratio_out, ratio.data(), (size_t)n_pairs * sizeof(float));
    std::sort(ratio.begin(), ratio.end());                                    // :654-662
    *median_out = ratio[n_pairs / 2];
    return RUMI_OK;
}

extern "C" int rumi_sim3_ransac(RumiOptimizer *o, int32_t n, const float *X3Dc1, const float *X3Dc2, const float *sigma2_1, const float *sigma2_2,
                                const float *K4_1, const float *K4_2, int32_t fix_scale, int32_t n_hyp, const int32_t *triples,
                                const RumiSim3ScoreSet *score, float *T12_out, int32_t *n_inliers_out, uint8_t *inlier_out, float *ratio_out,
                                float *median_out) {
    if (!o || n < 3 || n_hyp < 0 || !X3Dc1 || !X3Dc2 || !sigma2_1 || !sigma2_2 || !K4_1 || !K4_2 || !T12_out || !n_inliers_out) return RUMI_E_INVALID;
    if (n_hyp == 0) return RUMI_OK;
    if (!triples) return RUMI_E_INVALID;
    for (int i = 0; i < 3 * n_hyp; i++) if (triples[i] < 0 || triples[i] >= n) { g_lastError = "rumi_sim3_ransac: correspondence index out of range"; return RUMI_E_INVALID; }
    int total = 0, np = 0;
    if (score) {
        np = score->n_pairs;
        if (np < 1 || !score->pair_start || !score->pair_denominator || !score->S_c1w1 || !score->S_c2w2 || !score->S_kf1w || !score->S_kf2w || !score->K4_1 ||
            !score->K4_2 || !median_out) return RUMI_E_INVALID;
        total = score->pair_start[np];
        if (total < 0 || (total > 0 && (!score->X1 || !score->X2 || !score->kp1 || !score->kp2 || !score->sigma2_1 || !score->sigma2_2 || !score->edge1 || !score->edge2)))
            return RUMI_E_INVALID;
    }
    HIP_TRY(hipSetDevice(o->device));
    auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t N = (size_t)n, H = (size_t)n_hyp, T = (size_t)total, NP = (size_t)np;
    const size_t oX1 = 0, oX2 = al(oX1 + N * 12), oT1 = al(oX2 + N * 12), oT2 = al(oT1 + N * 4), oK = al(oT2 + N * 4), oTri = al(oK + 64), oKf = al(oTri + H * 12),
                 oA = al(oKf + 128), oB = al(oA + NP * 64), oP = al(oB + NP * 64), sX1 = al(oP + T * 4), sX2 = al(sX1 + T * 12), sK1 = al(sX2 + T * 12),
                 sK2 = al(sK1 + T * 8), sS1 = al(sK2 + T * 8), sS2 = al(sS1 + T * 4), sE1 = al(sS2 + T * 4), sE2 = al(sE1 + T), inBytes = al(sE2 + T);
    const size_t rT = 0, rN = al(rT + H * 64), rC = al(rN + H * 4), rI = al(rC + H * NP * 4), outBytes = al(rI + (inlier_out ? H * N : 0)),
                 rComp = outBytes, scratchEnd = al(rComp + H * NP * 128);
    if (inBytes > o->baStageCap || scratchEnd > o->baStageCap) { g_lastError = "rumi_sim3_ransac: more correspondences / hypotheses than the optimiser's arenas hold"; return RUMI_E_CAPACITY; }
    uint8_t *h = o->hBa;
    std::memcpy(h + oX1, X3Dc1, N * 12); std::memcpy(h + oX2, X3Dc2, N * 12);
    float *t1 = reinterpret_cast<float *>(h + oT1), *t2 = reinterpret_cast<float *>(h + oT2);
    for (int i = 0; i < n; i++) {        // mvnMaxError1/2 are vector<size_t> upstream (Sim3Solver.h:77-78): 9.210 * sigma2 truncated, compared as float
        t1[i] = (float)(size_t)(9.210 * (double)sigma2_1[i]);
        t2[i] = (float)(size_t)(9.210 * (double)sigma2_2[i]);
    }
    std::memcpy(h + oK, K4_1, 16); std::memcpy(h + oK + 16, K4_2, 16);
    std::memcpy(h + oTri, triples, H * 12);
    if (score) {
        std::memcpy(h + oK + 32, score->K4_1, 16); std::memcpy(h + oK + 48, score->K4_2, 16);
        std::memcpy(h + oKf, score->S_kf1w, 64); std::memcpy(h + oKf + 64, score->S_kf2w, 64);
        std::memcpy(h + oA, score->S_c1w1, NP * 64); std::memcpy(h + oB, score->S_c2w2, NP * 64);
        int32_t *pairOf = reinterpret_cast<int32_t *>(h + oP);
        for (int p = 0; p < np; p++) {
            if (score->pair_start[p] > score->pair_start[p + 1] || score->pair_start[p] < 0) { g_lastError = "rumi_sim3_ransac: pair_start is not ascending"; return RUMI_E_INVALID; }
            for (int i = score->pair_start[p]; i < score->pair_start[p + 1]; i++) pairOf[i] = p;
        }
        if (total) {
            std::memcpy(h + sX1, score->X1, T * 12); std::memcpy(h + sX2, score->X2, T * 12); std::memcpy(h + sK1, score->kp1, T * 8); std::memcpy(h + sK2, score->kp2, T * 8);
            std::memcpy(h + sS1, score->sigma2_1, T * 4); std::memcpy(h + sS2, score->sigma2_2, T * 4); std::memcpy(h + sE1, score->edge1, T); std::memcpy(h + sE2, score->edge2, T);
        }
    }
    HIP_TRY(hipMemcpyAsync(o->dBa, h, inBytes, hipMemcpyHostToDevice, nullptr));
    uint8_t *d = o->dBa, *r = o->dBaOut;
    RansacArgs A;
    A.n = n; A.nHyp = n_hyp; A.fixScale = fix_scale != 0;
    A.X1 = (const float *)(d + oX1); A.X2 = (const float *)(d + oX2); A.thr1 = (const float *)(d + oT1); A.thr2 = (const float *)(d + oT2);
    A.K1 = (const float *)(d + oK); A.K2 = (const float *)(d + oK + 16); A.tri = (const int32_t *)(d + oTri);
    A.T12 = (float *)(r + rT); A.nIn = (int32_t *)(r + rN); A.inl = inlier_out ? r + rI : nullptr;
    A.nPairs = np; A.total = score ? (total > 0 ? total : 0) : 0;
    A.pairOf = (const int32_t *)(d + oP); A.Sc1w1 = (const double *)(d + oA); A.Sc2w2 = (const double *)(d + oB); A.Skf = (const double *)(d + oKf);
    A.sK1 = (const float *)(d + oK + 32); A.sK2 = (const float *)(d + oK + 48); A.sX1 = (const float *)(d + sX1); A.sX2 = (const float *)(d + sX2);
    A.kp1 = (const float *)(d + sK1); A.kp2 = (const float *)(d + sK2); A.sg1 = (const float *)(d + sS1); A.sg2 = (const float *)(d + sS2);
    A.e1 = d + sE1; A.e2 = d + sE2; A.pairCnt = (int32_t *)(r + rC); A.comp = (double *)(r + rComp);
    hipLaunchKernelGGL(k_sim3_ransac, dim3(n_hyp), dim3(256), 0, nullptr, A);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(o->hBa, r, outBytes, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    std::memcpy(T12_out, o->hBa + rT, H * 64);
    std::memcpy(n_inliers_out, o->hBa + rN, H * 4);
    if (inlier_out) std::memcpy(inlier_out, o->hBa + rI, H * N);
    if (score) {                                                              // :643-662 per hypothesis
        const int32_t *cnt = reinterpret_cast<const int32_t *>(o->hBa + rC);
        std::vector<float> ratio(np);
        for (int hh = 0; hh < n_hyp; hh++) {
            for (int p = 0; p < np; p++) ratio[p] = (total > 0 && score->pair_denominator[p]) ? (float)cnt[(size_t)hh * np + p] / (float)score->pair_denominator[p] : 0.f;
            if (ratio_out) std::memcpy(ratio_out + (size_t)hh * np, ratio.data(), NP * sizeof(float));
            std::sort(ratio.begin(), ratio.end());
            median_out[hh] = ratio[np / 2];
        }
    }
    return RUMI_OK;
}

extern "C" int rumi_optimize_sim3(RumiOptimizer *o, int32_t n, const int32_t *pair_of, int32_t n_pairs, const double *S_c1w, const double *S_c2w,
                                  const float *P1c, const float *P2c, const float *obs1, const float *obs2, const float *inv_sigma2_1,
                                  const float *inv_sigma2_2, const uint8_t *skip12, const uint8_t *skip21, const float *K4_1, const float *K4_2,
                                  float th2, int32_t fix_scale, int32_t robust_first_pass, double *S_io8, uint8_t *status_out, int32_t *result3) {
    if (!o || n < 0 || !K4_1 || !K4_2 || !S_io8 || !result3) return RUMI_E_INVALID;
    if (n > 0 && (!P1c || !P2c || !obs1 || !obs2 || !inv_sigma2_1 || !inv_sigma2_2 || !status_out)) return RUMI_E_INVALID;
    const bool world = S_c1w != nullptr;
    if (world && (!S_c2w || n_pairs < 1 || (n > 0 && !pair_of))) return RUMI_E_INVALID;
    if (!world) n_pairs = 1;
    if (world) for (int i = 0; i < n; i++) if (pair_of[i] < 0 || pair_of[i] >= n_pairs) { g_lastError = "rumi_optimize_sim3: pair index out of range"; return RUMI_E_INVALID; }
    HIP_TRY(hipSetDevice(o->device));
    auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t N = (size_t)n, NP = (size_t)n_pairs;
    const size_t oS = 0, oA = al(oS + 64), oB = al(oA + NP * 64), oK = al(oB + NP * 64), oP = al(oK + 32), oX1 = al(oP + N * 4), oX2 = al(oX1 + N * 12),
                 oO1 = al(oX2 + N * 12), oO2 = al(oO1 + N * 8), oW1 = al(oO2 + N * 8), oW2 = al(oW1 + N * 4), oE1 = al(oW2 + N * 4), oE2 = al(oE1 + N),
                 inBytes = al(oE2 + N);
    // result block: estimate, counters, status; scratch behind it
    const size_t rS = 0, rR = 64, rSt = 80, outBytes = al(rSt + N), sC = outBytes, sX1 = al(sC + NP * 30 * 64), sX2 = al(sX1 + N * 8), sO1 = al(sX2 + N * 8),
                 sO2 = al(sO1 + N), scratchEnd = al(sO2 + N);
    if (inBytes > o->baStageCap || scratchEnd > o->baStageCap) { g_lastError = "rumi_optimize_sim3: more correspondences than the optimiser's arenas hold"; return RUMI_E_CAPACITY; }
    uint8_t *h = o->hBa;
    std::memcpy(h + oS, S_io8, 64);
    if (world) { std::memcpy(h + oA, S_c1w, NP * 64); std::memcpy(h + oB, S_c2w, NP * 64); if (n) std::memcpy(h + oP, pair_of, N * 4); }
    std::memcpy(h + oK, K4_1, 16); std::memcpy(h + oK + 16, K4_2, 16);
    if (n) {
        std::memcpy(h + oX1, P1c, N * 12); std::memcpy(h + oX2, P2c, N * 12); std::memcpy(h + oO1, obs1, N * 8); std::memcpy(h + oO2, obs2, N * 8);
        std::memcpy(h + oW1, inv_sigma2_1, N * 4); std::memcpy(h + oW2, inv_sigma2_2, N * 4);
        if (skip12) std::memcpy(h + oE1, skip12, N); else std::memset(h + oE1, 0, N);
        if (skip21) std::memcpy(h + oE2, skip21, N); else std::memset(h + oE2, 0, N);
    }
    HIP_TRY(hipMemcpyAsync(o->dBa, h, inBytes, hipMemcpyHostToDevice, nullptr));
    uint8_t *d = o->dBa, *r = o->dBaOut;
    Sim3Args A;
    A.n = n; A.nPairs = n_pairs; A.world = world; A.fixScale = fix_scale != 0; A.robustFirst = robust_first_pass != 0; A.th2 = th2;
    A.pairOf = world ? (const int32_t *)(d + oP) : nullptr;
    A.Sc1w = (const double *)(d + oA); A.Sc2w = (const double *)(d + oB); A.Sin = (const double *)(d + oS);
    A.P1c = (const float *)(d + oX1); A.P2c = (const float *)(d + oX2); A.obs1 = (const float *)(d + oO1); A.obs2 = (const float *)(d + oO2);
    A.w1 = (const float *)(d + oW1); A.w2 = (const float *)(d + oW2); A.skip12 = d + oE1; A.skip21 = d + oE2;
    A.K1 = (const float *)(d + oK); A.K2 = (const float *)(d + oK + 16);
    A.Sout = (double *)(r + rS); A.res = (int32_t *)(r + rR); A.status = r + rSt;
    A.comp = (double *)(r + sC); A.chi12 = (double *)(r + sX1); A.chi21 = (double *)(r + sX2); A.on12 = r + sO1; A.on21 = r + sO2;
    hipLaunchKernelGGL(k_sim3_opt, dim3(1), dim3(256), 0, nullptr, A);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(o->hBa, r, outBytes, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    std::memcpy(S_io8, o->hBa + rS, 64);
    std::memcpy(result3, o->hBa + rR, 12);
    if (n) std::memcpy(status_out, o->hBa + rSt, N);
    return RUMI_OK;
}
