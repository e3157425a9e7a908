// Facade for Sim3Solver::ComputeInliersNum (R/lib_src/Sim3Solver.cc:564-664; declared static in Sim3Solver.h), the alignment score
// CloudMerging::ComputeSubmapSim3 evaluates for every Sim3 hypothesis (CloudMerging.cc:611,748,809).  The composed transforms are
// formed here with the caller's own g2o::Sim3 (same expressions as :598-599, :620-621); the per-match re-projections and tests run
// on the GPU (include/rumi_opt.h: rumi_sim3_inliers).  Templated so that it instantiates against the reference's g2o::Sim3 /
// KeyFrame / MapPoint and against the stand-ins of tests/cpp.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

#include "Optimizer.h"
#include "rumi_status.h"

namespace rumi_facade {

template <class KeyFrameT, class Sim3T>
float ComputeInliersNum(const std::vector<KeyFrameT *> &map1KFs, const std::vector<KeyFrameT *> &map2KFs,
                        const std::vector<std::vector<std::pair<int, int>>> &avpValidKPMatches, Sim3T &gSw1w2) {
    std::vector<int32_t> pairStart(1, 0), denom;
    std::vector<double> A, B;
    std::vector<float> X1, X2, k1, k2, s1, s2;
    std::vector<uint8_t> e1, e2;
    float K1[4] = {0, 0, 0, 0}, K2[4] = {0, 0, 0, 0};
    auto push_sim3 = [](std::vector<double> &v, const Sim3T &S) {
        const auto q = S.rotation(); const auto t = S.translation();
        const double a[8] = {q.x(), q.y(), q.z(), q.w(), t(0), t(1), t(2), S.scale()};
        v.insert(v.end(), a, a + 8);
    };
    for (size_t i = 0; i < map1KFs.size(); i++) {
        KeyFrameT *KF1 = map1KFs[i], *KF2 = map2KFs[i];
        if (!KF1 || !KF2) continue;                                                           // :589-592
        const auto Tc1w1 = KF1->GetPose(); const auto Tc2w2 = KF2->GetPose();
        const Sim3T gSc1w1(Tc1w1.rotationMatrix().template cast<double>(), Tc1w1.translation().template cast<double>(), 1.0);
        const Sim3T gSc2w2(Tc2w2.rotationMatrix().template cast<double>(), Tc2w2.translation().template cast<double>(), 1.0);
        const Sim3T gSc2w1 = gSc2w2 * gSw1w2.inverse(), gSc1w2 = gSc1w1 * gSw1w2;            // :620-621
        push_sim3(A, gSc1w2); push_sim3(B, gSc2w1);
        K1[0] = KF1->fx; K1[1] = KF1->fy; K1[2] = KF1->cx; K1[3] = KF1->cy;
        K2[0] = KF2->fx; K2[1] = KF2->fy; K2[2] = KF2->cx; K2[3] = KF2->cy;
        const auto vpMPs1 = KF1->GetMapPointMatches(), vpMPs2 = KF2->GetMapPointMatches();
        for (const auto &m : avpValidKPMatches[i]) {
            auto *p1 = vpMPs1[m.first]; auto *p2 = vpMPs2[m.second];
            if (!p1 || !p2) continue;                                                         // :611-614 (still counted in the ratio's denominator)
            const auto P1 = p1->GetWorldPos(), P2 = p2->GetWorldPos();
            for (int c = 0; c < 3; c++) { X1.push_back(P1(c)); X2.push_back(P2(c)); }
            const auto &kp1 = KF1->mvKeys[m.first]; const auto &kp2 = KF2->mvKeys[m.second];
            k1.push_back(kp1.pt.x); k1.push_back(kp1.pt.y); k2.push_back(kp2.pt.x); k2.push_back(kp2.pt.y);
            s1.push_back(KF1->mvLevelSigma2[kp1.octave]); s2.push_back(KF2->mvLevelSigma2[kp2.octave]);
            e1.push_back(p1->isEdge); e2.push_back(p2->isEdge);
        }
        pairStart.push_back((int32_t)e1.size());
        denom.push_back((int32_t)avpValidKPMatches[i].size());
    }
    const int nPairs = (int)denom.size();
    if (nPairs == 0) return 0.f;
    std::vector<uint8_t> inl(e1.size() + 1);
    float median = 0.f;
    if (RUMI_GUARDED("Sim3Scoring::ComputeInliersNum / rumi_sim3_inliers", &RUMI_FACADE_NAMESPACE::Optimizer::grow_arena,
                     rumi_sim3_inliers(RUMI_FACADE_NAMESPACE::Optimizer::arena(), nPairs, pairStart.data(), denom.data(), A.data(), B.data(), K1, K2, X1.data(), X2.data(),
                                       k1.data(), k2.data(), s1.data(), s2.data(), e1.data(), e2.data(), inl.data(), nullptr, &median)) != RUMI_OK)
        return 0.f;                                         // reported (rumi_status.h); an alignment with no inliers
    return median;
}

}  // namespace rumi_facade
