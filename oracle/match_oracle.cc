// ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's Hamming matchers on flat arrays
// (map points / key-frames are integer ids; a NULL MapPoint* is -1).  R/ = /root/reference/src/rumi-slam/.
//   DescriptorDistance                              R/lib_src/ORBmatcher.cc:1830-1844
//   Frame::AssignFeaturesToGrid / PosInGrid         R/lib_src/Frame.cc:441-466, 752-761
//   Frame::GetFeaturesInArea                        R/lib_src/Frame.cc:695-750
//   SearchByProjection(Frame&, vector<MapPoint*>&)  R/lib_src/ORBmatcher.cc:39-196        (mono branch)
//   SearchByBoW(KeyFrame*, Frame&, ...)             R/lib_src/ORBmatcher.cc:198-370       (mono branch)
//   SearchByProjection(Frame& Cur, const Frame& Last) R/lib_src/ORBmatcher.cc:1498-1683   (mono branch)
//   SearchByProjection(KeyFrame*, Sim3f&, points[, pointKFs], matched[, matchedKF], th, ratioHamming)  :372-471, :473-579
//   SearchByProjection(Frame&, KeyFrame*, set<MapPoint*>&, th, ORBdist)   R/lib_src/ORBmatcher.cc:1685-1793
//   SearchByBoW(KeyFrame*, KeyFrame*, ...)          R/lib_src/ORBmatcher.cc:682-804
//   KeyFrame::GetFeaturesInArea / IsInImage         R/lib_src/KeyFrame.cc:887-929;  MapPoint::PredictScale  R/lib_src/MapPoint.cc:538-570
//   ComputeThreeMaxima                              R/lib_src/ORBmatcher.cc:1795-1826
//   Sophus SE3f * point, Pinhole::project           R/Thirdparty/Sophus/sophus/so3.hpp:358-367, R/lib_src/CameraModels/Pinhole.cpp:43-49
// PARITY STATUS: pinned by source only (the reference has no tests for this path); integer logic is exact,
// the float projection follows Eigen's expression order (no FMA: build with -ffp-contract=off).
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "orb_oracle.h"

namespace orc {

static const int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;   // ORBmatcher.cc:30-32
static const int GRID_COLS = 64, GRID_ROWS = 48;                  // Frame.h:42-43

int descriptor_distance(const uint8_t *a, const uint8_t *b) {
    int32_t pa[8], pb[8];
    std::memcpy(pa, a, 32); std::memcpy(pb, b, 32);
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        unsigned int v = pa[i] ^ pb[i];
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

struct FrameGrid {
    int n;
    const KeyPoint *keys;        // mvKeysUn
    const uint8_t *desc;         // mDescriptors
    float minX, minY, maxX, maxY, wInv, hInv;
    std::vector<int> cell[GRID_COLS][GRID_ROWS];

    FrameGrid(const KeyPoint *k, const uint8_t *d, int n_, float minX_, float minY_, float maxX_, float maxY_)
        : n(n_), keys(k), desc(d), minX(minX_), minY(minY_), maxX(maxX_), maxY(maxY_) {
        wInv = (float)GRID_COLS / (float)(maxX - minX);      // Frame.cc:322-323
        hInv = (float)GRID_ROWS / (float)(maxY - minY);
        for (int i = 0; i < n; i++) {                          // AssignFeaturesToGrid + PosInGrid
            int px = (int)std::round((keys[i].x - minX) * wInv);
            int py = (int)std::round((keys[i].y - minY) * hInv);
            if (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) continue;
            cell[px][py].push_back(i);
        }
    }
    // GetFeaturesInArea(x, y, r, minLevel, maxLevel)
    void in_area(float x, float y, float r, int minLevel, int maxLevel, std::vector<int> &out) const {
        out.clear();
        const int nMinCellX = std::max(0, (int)std::floor((x - minX - r) * wInv));
        if (nMinCellX >= GRID_COLS) return;
        const int nMaxCellX = std::min(GRID_COLS - 1, (int)std::ceil((x - minX + r) * wInv));
        if (nMaxCellX < 0) return;
        const int nMinCellY = std::max(0, (int)std::floor((y - minY - r) * hInv));
        if (nMinCellY >= GRID_ROWS) return;
        const int nMaxCellY = std::min(GRID_ROWS - 1, (int)std::ceil((y - minY + r) * hInv));
        if (nMaxCellY < 0) return;
        const bool checkLevels = (minLevel > 0) || (maxLevel >= 0);
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++)
                for (int idx : cell[ix][iy]) {
                    const KeyPoint &kp = keys[idx];
                    if (checkLevels) {
                        if (kp.octave < minLevel) continue;
                        if (maxLevel >= 0 && kp.octave > maxLevel) continue;
                    }
                    const float dx = kp.x - x, dy = kp.y - y;
                    if (std::fabs(dx) < r && std::fabs(dy) < r) out.push_back(idx);
                }
    }
};

static void three_maxima(const std::vector<int> *histo, int L, int &ind1, int &ind2, int &ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = (int)histo[i].size();
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) ind3 = -1;
}

static int rot_bin(float angleA, float angleB) {
    const float factor = 1.0f / HISTO_LENGTH;
    float rot = angleA - angleB;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)std::round(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

// Sophus::SE3f * Vector3f with q = (x,y,z,w), t:  p + w*uv + q.vec x uv  (uv = 2 * q.vec x p), then + t
static void se3_mul(const float *T /*qx qy qz qw tx ty tz*/, const float *p, float *o) {
    const float qx = T[0], qy = T[1], qz = T[2], qw = T[3];
    float uv[3] = {qy * p[2] - qz * p[1], qz * p[0] - qx * p[2], qx * p[1] - qy * p[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    const float c[3] = {qy * uv[2] - qz * uv[1], qz * uv[0] - qx * uv[2], qx * uv[1] - qy * uv[0]};
    for (int i = 0; i < 3; i++) o[i] = ((p[i] + qw * uv[i]) + c[i]) + T[4 + i];
}

}  // namespace orc

using namespace orc;

extern "C" {

int orc_descriptor_distance(const uint8_t *a, const uint8_t *b) { return descriptor_distance(a, b); }

// GetFeaturesInArea on a frame; returns the number of indices written (enumeration order).
int orc_features_in_area(const KeyPoint *keys, int n, float minX, float minY, float maxX, float maxY, float x, float y,
                         float r, int minLevel, int maxLevel, int32_t *out, int cap) {
    static const uint8_t dummy[32] = {0};
    FrameGrid g(keys, dummy, n, minX, minY, maxX, maxY);
    std::vector<int> v;
    g.in_area(x, y, r, minLevel, maxLevel, v);
    for (int i = 0; i < (int)v.size() && i < cap; i++) out[i] = v[i];
    return (int)v.size();
}

// SearchByProjection(Frame &F, const vector<MapPoint*>&, th, bFarPoints, thFarPoints), mono.
// frame_mp[n]: map point id per feature (-1 = none), updated in place.  Returns nmatches.
int orc_search_by_projection_mappoints(const KeyPoint *keys, const uint8_t *desc, int n, float minX, float minY, float maxX,
                                       float maxY, const float *scaleFactors, int nmp, const uint8_t *trackInView,
                                       const float *projX, const float *projY, const int32_t *scaleLevel,
                                       const float *viewCos, const float *trackDepth, const uint8_t *isBad,
                                       const uint8_t *mpDesc, const int32_t *mpObs, float th, int bFarPoints,
                                       float thFarPoints, float nnratio, int32_t *frame_mp) {
    FrameGrid F(keys, desc, n, minX, minY, maxX, maxY);
    int nmatches = 0;
    const bool bFactor = th != 1.0;
    std::vector<int> vIndices;
    for (int iMP = 0; iMP < nmp; iMP++) {
        if (!trackInView[iMP]) continue;
        if (bFarPoints && trackDepth[iMP] > thFarPoints) continue;
        if (isBad[iMP]) continue;
        const int nPredictedLevel = scaleLevel[iMP];
        float r = viewCos[iMP] > 0.998 ? 2.5f : 4.0f;     // RadiusByViewingCos
        if (bFactor) r *= th;
        F.in_area(projX[iMP], projY[iMP], r * scaleFactors[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel, vIndices);
        if (vIndices.empty()) continue;
        const uint8_t *MPd = mpDesc + (size_t)iMP * 32;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int idx : vIndices) {
            if (frame_mp[idx] >= 0 && mpObs[frame_mp[idx]] > 0) continue;
            const int dist = descriptor_distance(MPd, desc + (size_t)idx * 32);
            if (dist < bestDist) {
                bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = keys[idx].octave; bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = keys[idx].octave; bestDist2 = dist;
            }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            if (bestLevel != bestLevel2 || bestDist <= nnratio * bestDist2) {
                frame_mp[bestIdx] = iMP;
                nmatches++;
            }
        }
    }
    return nmatches;
}

// SearchByProjection(Frame &Cur, const Frame &Last, th, bMono=true), mono.
// last_mp[nlast]: map point id of each last-frame feature (-1 none); cur_mp[ncur] in/out.
int orc_search_by_projection_frame(const KeyPoint *curKeys, const uint8_t *curDesc, int ncur, float minX, float minY,
                                   float maxX, float maxY, const float *scaleFactors, const float *Tcw7, const float *K4,
                                   const KeyPoint *lastKeys, int nlast, const int32_t *lastMp, const uint8_t *lastOutlier,
                                   const float *mpPos, const uint8_t *mpDesc, const int32_t *mpObs, float th, int checkOri,
                                   int32_t *cur_mp) {
    FrameGrid C(curKeys, curDesc, ncur, minX, minY, maxX, maxY);
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    std::vector<int> vIndices2;
    for (int i = 0; i < nlast; i++) {
        const int pMP = lastMp[i];
        if (pMP < 0 || lastOutlier[i]) continue;
        float x3Dc[3];
        se3_mul(Tcw7, mpPos + (size_t)pMP * 3, x3Dc);
        const float invzc = (float)(1.0 / x3Dc[2]);
        if (invzc < 0) continue;
        const float u = K4[0] * x3Dc[0] / x3Dc[2] + K4[2], v = K4[1] * x3Dc[1] / x3Dc[2] + K4[3];
        if (u < minX || u > maxX) continue;
        if (v < minY || v > maxY) continue;
        const int nLastOctave = lastKeys[i].octave;
        const float radius = th * scaleFactors[nLastOctave];
        C.in_area(u, v, radius, nLastOctave - 1, nLastOctave + 1, vIndices2);   // mono: neither forward nor backward
        if (vIndices2.empty()) continue;
        const uint8_t *dMP = mpDesc + (size_t)pMP * 32;
        int bestDist = 256, bestIdx2 = -1;
        for (int i2 : vIndices2) {
            if (cur_mp[i2] >= 0 && mpObs[cur_mp[i2]] > 0) continue;
            const int dist = descriptor_distance(dMP, curDesc + (size_t)i2 * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= TH_HIGH) {
            cur_mp[bestIdx2] = pMP;
            nmatches++;
            if (checkOri) rotHist[rot_bin(lastKeys[i].angle, curKeys[bestIdx2].angle)].push_back(bestIdx2);
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int idx : rotHist[i]) { cur_mp[idx] = -1; nmatches--; }
    }
    return nmatches;
}

// SearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches), mono.
// FeatureVectors in CSR form: node ids ascending (std::map order), offsets[nn+1], feature indices.
int orc_search_by_bow(const KeyPoint *kfKeys, const uint8_t *kfDesc, int nkf, const int32_t *kfMp, const uint8_t *mpBad,
                      const uint32_t *kfNodes, const int32_t *kfOff, const uint32_t *kfIdx, int nnKF,
                      const KeyPoint *fKeys, const uint8_t *fDesc, int nf, const uint32_t *fNodes, const int32_t *fOff,
                      const uint32_t *fIdx, int nnF, float nnratio, int checkOri, int32_t *matches /*[nf]*/) {
    for (int i = 0; i < nf; i++) matches[i] = -1;
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    int a = 0, b = 0;
    while (a < nnKF && b < nnF) {
        if (kfNodes[a] == fNodes[b]) {
            for (int p = kfOff[a]; p < kfOff[a + 1]; p++) {
                const unsigned realIdxKF = kfIdx[p];
                const int pMP = kfMp[realIdxKF];
                if (pMP < 0) continue;
                if (mpBad[pMP]) continue;
                const uint8_t *dKF = kfDesc + (size_t)realIdxKF * 32;
                int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
                for (int q = fOff[b]; q < fOff[b + 1]; q++) {
                    const unsigned realIdxF = fIdx[q];
                    if (matches[realIdxF] >= 0) continue;
                    const int dist = descriptor_distance(dKF, fDesc + (size_t)realIdxF * 32);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = (int)realIdxF; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 <= TH_LOW) {
                    if ((float)bestDist1 < nnratio * (float)bestDist2) {
                        matches[bestIdxF] = pMP;
                        if (checkOri) rotHist[rot_bin(kfKeys[realIdxKF].angle, fKeys[bestIdxF].angle)].push_back(bestIdxF);
                        nmatches++;
                    }
                }
            }
            a++; b++;
        } else if (kfNodes[a] < fNodes[b]) {
            a = (int)(std::lower_bound(kfNodes, kfNodes + nnKF, fNodes[b]) - kfNodes);
        } else {
            b = (int)(std::lower_bound(fNodes, fNodes + nnF, kfNodes[a]) - fNodes);
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx : rotHist[i]) { matches[idx] = -1; nmatches--; }
        }
    }
    return nmatches;
}

// MapPoint::PredictScale (R/lib_src/MapPoint.cc:538-570): ceil(log(mfMaxDistance / dist) / mfLogScaleFactor), clamped.
// `log` is called unqualified on a float in a file without `using namespace std`: it binds to ::log(double) unless a header
// pulls libstdc++'s <math.h> overloads in; the double form is restated here (they differ only when the quotient is within
// one float ulp of an integer) — unpinned.
static int predict_scale(float maxDistance, float dist, float logScaleFactor, int nLevels) {
    const float ratio = maxDistance / dist;
    int nScale = (int)std::ceil(std::log((double)ratio) / logScaleFactor);
    if (nScale < 0) nScale = 0;
    else if (nScale >= nLevels) nScale = nLevels - 1;
    return nScale;
}

// SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12), mono  (ORBmatcher.cc:682-804).
// matches12[n1]: index of the KF2 FEATURE matched to each KF1 feature (-1 none); the caller maps it to vpMapPoints2[.].
int orc_search_by_bow_kf(const KeyPoint *k1, const uint8_t *d1, int n1, const int32_t *mp1, const uint32_t *nodes1, const int32_t *off1,
                         const uint32_t *idx1, int nn1, const KeyPoint *k2, const uint8_t *d2, int n2, const int32_t *mp2,
                         const uint32_t *nodes2, const int32_t *off2, const uint32_t *idx2, int nn2, const uint8_t *mpBad,
                         float nnratio, int checkOri, int32_t *matches12) {
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    std::vector<char> matched2(n2, 0);
    std::vector<int> rotHist[HISTO_LENGTH];
    int nmatches = 0, a = 0, b = 0;
    while (a < nn1 && b < nn2) {
        if (nodes1[a] == nodes2[b]) {
            for (int p = off1[a]; p < off1[a + 1]; p++) {
                const int i1 = (int)idx1[p];
                const int pMP1 = mp1[i1];
                if (pMP1 < 0 || mpBad[pMP1]) continue;
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (int q = off2[b]; q < off2[b + 1]; q++) {
                    const int i2 = (int)idx2[q];
                    const int pMP2 = mp2[i2];
                    if (matched2[i2] || pMP2 < 0) continue;
                    if (mpBad[pMP2]) continue;
                    const int dist = descriptor_distance(d1 + (size_t)i1 * 32, d2 + (size_t)i2 * 32);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = i2; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 < TH_LOW && (float)bestDist1 < nnratio * (float)bestDist2) {
                    matches12[i1] = bestIdx2;
                    matched2[bestIdx2] = 1;
                    if (checkOri) rotHist[rot_bin(k1[i1].angle, k2[bestIdx2].angle)].push_back(i1);
                    nmatches++;
                }
            }
            a++; b++;
        } else if (nodes1[a] < nodes2[b]) a = (int)(std::lower_bound(nodes1, nodes1 + nn1, nodes2[b]) - nodes1);
        else b = (int)(std::lower_bound(nodes2, nodes2 + nn2, nodes1[a]) - nodes2);
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx : rotHist[i]) { matches12[idx] = -1; nmatches--; }
        }
    }
    return nmatches;
}

// SearchByProjection(KeyFrame *pKF, Sim3f &Scw, vpPoints, [vpPointsKFs,] vpMatched, [vpMatchedKF,] th, ratioHamming)
// (ORBmatcher.cc:372-471 with variant = 0: camera->project; :473-579 with variant = 1: explicit invz form).
// Tcw7 / Ow3 are the SE3 and camera centre the reference derives from Scw (:380-381); skip[i] = isBad() || spAlreadyFound.count().
// matched[nkf] in/out: index of the POINT (into vpPoints) matched to each key-frame feature, -2 = already held a point, -1 = free.
int orc_search_by_projection_sim3(const KeyPoint *keys, const uint8_t *desc, int n, float minX, float minY, float maxX, float maxY,
                                  const float *scaleFactors, int nLevels, float logScaleFactor, const float *Tcw7, const float *Ow3,
                                  const float *K4, int nmp, const uint8_t *skip, const float *mpPos, const float *mpNormal,
                                  const float *mpMinDist, const float *mpMaxDist, const uint8_t *mpDesc, int th, float ratioHamming,
                                  int variant, int32_t *matched) {
    FrameGrid KF(keys, desc, n, minX, minY, maxX, maxY);
    int nmatches = 0;
    std::vector<int> vIndices;
    for (int iMP = 0; iMP < nmp; iMP++) {
        if (skip[iMP]) continue;
        const float *p3Dw = mpPos + (size_t)iMP * 3;
        float p3Dc[3];
        se3_mul(Tcw7, p3Dw, p3Dc);
        if (p3Dc[2] < 0.0) continue;
        float u, v;
        if (variant == 0) { u = K4[0] * p3Dc[0] / p3Dc[2] + K4[2]; v = K4[1] * p3Dc[1] / p3Dc[2] + K4[3]; }
        else { const float invz = 1 / p3Dc[2]; const float x = p3Dc[0] * invz, y = p3Dc[1] * invz; u = K4[0] * x + K4[2]; v = K4[1] * y + K4[3]; }
        if (!(u >= minX && u < maxX && v >= minY && v < maxY)) continue;                 // KeyFrame::IsInImage
        const float maxDistance = 1.2f * mpMaxDist[iMP], minDistance = 0.8f * mpMinDist[iMP];
        const float PO[3] = {p3Dw[0] - Ow3[0], p3Dw[1] - Ow3[1], p3Dw[2] - Ow3[2]};
        const float dist = std::sqrt((PO[0] * PO[0] + PO[1] * PO[1]) + PO[2] * PO[2]);
        if (dist < minDistance || dist > maxDistance) continue;
        const float *Pn = mpNormal + (size_t)iMP * 3;
        if ((PO[0] * Pn[0] + PO[1] * Pn[1]) + PO[2] * Pn[2] < 0.5 * dist) continue;        // viewing angle < 60 deg
        const int nPredictedLevel = predict_scale(mpMaxDist[iMP], dist, logScaleFactor, nLevels);
        const float radius = th * scaleFactors[nPredictedLevel];
        KF.in_area(u, v, radius, -1, -1, vIndices);                                         // KeyFrame::GetFeaturesInArea: no level filter
        if (vIndices.empty()) continue;
        const uint8_t *dMP = mpDesc + (size_t)iMP * 32;
        int bestDist = 256, bestIdx = -1;
        for (int idx : vIndices) {
            if (matched[idx] != -1) continue;
            const int kpLevel = keys[idx].octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            const int d = descriptor_distance(dMP, desc + (size_t)idx * 32);
            if (d < bestDist) { bestDist = d; bestIdx = idx; }
        }
        if (bestDist <= TH_LOW * ratioHamming) { matched[bestIdx] = iMP; nmatches++; }
    }
    return nmatches;
}

// The search half of ORBmatcher::Fuse (R/lib_src/ORBmatcher.cc:1058-1161 with checkReproj = 1, monocular key-points;
// :1209-1277 with checkReproj = 0): per map point the key-frame feature of smallest Hamming distance inside the predicted-scale
// window, -1 when the point is not visible or the distance exceeds TH_LOW.  What follows in the reference (Replace /
// AddObservation / AddMapPoint and the isBad / IsInKeyFrame skips) acts on live map objects and is not restated here.
void orc_fuse_candidates(const KeyPoint *keys, const uint8_t *desc, int n, float minX, float minY, float maxX, float maxY,
                         const float *scaleFactors, int nLevels, float logScaleFactor, const float *Tcw7, const float *Ow3, const float *K4,
                         int nmp, const uint8_t *skip, const float *mpPos, const float *mpNormal, const float *mpMinDist,
                         const float *mpMaxDist, const uint8_t *mpDesc, float th, int checkReproj, int32_t *bestIdxOut) {
    FrameGrid KF(keys, desc, n, minX, minY, maxX, maxY);
    std::vector<int> vIndices;
    for (int i = 0; i < nmp; i++) {
        bestIdxOut[i] = -1;
        if (skip[i]) continue;
        const float *p3Dw = mpPos + (size_t)i * 3;
        float p3Dc[3];
        se3_mul(Tcw7, p3Dw, p3Dc);
        if (p3Dc[2] < 0.0f) continue;
        const float u = K4[0] * p3Dc[0] / p3Dc[2] + K4[2], v = K4[1] * p3Dc[1] / p3Dc[2] + K4[3];
        if (!(u >= minX && u < maxX && v >= minY && v < maxY)) continue;
        const float maxDistance = 1.2f * mpMaxDist[i], minDistance = 0.8f * mpMinDist[i];
        const float PO[3] = {p3Dw[0] - Ow3[0], p3Dw[1] - Ow3[1], p3Dw[2] - Ow3[2]};
        const float dist3D = std::sqrt((PO[0] * PO[0] + PO[1] * PO[1]) + PO[2] * PO[2]);
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const float *Pn = mpNormal + (size_t)i * 3;
        if ((PO[0] * Pn[0] + PO[1] * Pn[1]) + PO[2] * Pn[2] < 0.5 * dist3D) continue;
        const int nPredictedLevel = predict_scale(mpMaxDist[i], dist3D, logScaleFactor, nLevels);
        const float radius = th * scaleFactors[nPredictedLevel];
        KF.in_area(u, v, radius, -1, -1, vIndices);
        if (vIndices.empty()) continue;
        const uint8_t *dMP = mpDesc + (size_t)i * 32;
        int bestDist = 256, bestIdx = -1;
        for (int idx : vIndices) {
            const KeyPoint &kp = keys[idx];
            const int kpLevel = kp.octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            if (checkReproj) {
                const float ex = u - kp.x, ey = v - kp.y;
                const float e2 = ex * ex + ey * ey;
                const float invSigma2 = 1.0f / (scaleFactors[kpLevel] * scaleFactors[kpLevel]);       // ORBextractor.cc:417-426
                if (e2 * invSigma2 > 5.99) continue;
            }
            const int d = descriptor_distance(dMP, desc + (size_t)idx * 32);
            if (d < bestDist) { bestDist = d; bestIdx = idx; }
        }
        if (bestDist <= TH_LOW) bestIdxOut[i] = bestIdx;
    }
}

// One direction of ORBmatcher::SearchBySim3 (R/lib_src/ORBmatcher.cc:1329-1402 / :1405-1478) from camera-frame points
static void sim3_direction(const KeyPoint *keys, const uint8_t *desc, int n, float minX, float minY, float maxX, float maxY,
                           const float *scaleFactors, int nLevels, float logScaleFactor, const float *K4, int np, const uint8_t *skip,
                           const float *pc, const float *mpMinDist, const float *mpMaxDist, const uint8_t *mpDesc, float th,
                           std::vector<int> &vnMatch) {
    FrameGrid KF(keys, desc, n, minX, minY, maxX, maxY);
    vnMatch.assign(np, -1);
    std::vector<int> vIndices;
    for (int i = 0; i < np; i++) {
        if (skip[i]) continue;
        const float *p = pc + (size_t)i * 3;
        if (p[2] < 0.0) continue;
        const float invz = 1.0 / p[2];
        const float x = p[0] * invz, y = p[1] * invz;
        const float u = K4[0] * x + K4[2], v = K4[1] * y + K4[3];
        if (!(u >= minX && u < maxX && v >= minY && v < maxY)) continue;
        const float maxDistance = 1.2f * mpMaxDist[i], minDistance = 0.8f * mpMinDist[i];
        const float dist3D = std::sqrt((p[0] * p[0] + p[1] * p[1]) + p[2] * p[2]);
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const int nPredictedLevel = predict_scale(mpMaxDist[i], dist3D, logScaleFactor, nLevels);
        const float radius = th * scaleFactors[nPredictedLevel];
        KF.in_area(u, v, radius, -1, -1, vIndices);
        if (vIndices.empty()) continue;
        int bestDist = INT_MAX, bestIdx = -1;
        for (int idx : vIndices) {
            const int oct = keys[idx].octave;
            if (oct < nPredictedLevel - 1 || oct > nPredictedLevel) continue;
            const int d = descriptor_distance(mpDesc + (size_t)i * 32, desc + (size_t)idx * 32);
            if (d < bestDist) { bestDist = d; bestIdx = idx; }
        }
        if (bestDist <= TH_HIGH) vnMatch[i] = bestIdx;
    }
}

// ORBmatcher::SearchBySim3 (R/lib_src/ORBmatcher.cc:1293-1496); per-feature inputs as documented in include/rumi_match.h
int orc_search_by_sim3(const KeyPoint *keys1, const uint8_t *kfDesc1, int n1, const KeyPoint *keys2, const uint8_t *kfDesc2, int n2, float minX,
                       float minY, float maxX, float maxY, const float *scaleFactors, int nLevels, float logScaleFactor, const float *K4,
                       const uint8_t *skip1, const float *pc1in2, const float *min1, const float *max1, const uint8_t *desc1,
                       const uint8_t *skip2, const float *pc2in1, const float *min2, const float *max2, const uint8_t *desc2, float th,
                       int32_t *match12) {
    std::vector<int> vnMatch1, vnMatch2;
    sim3_direction(keys2, kfDesc2, n2, minX, minY, maxX, maxY, scaleFactors, nLevels, logScaleFactor, K4, n1, skip1, pc1in2, min1, max1, desc1, th, vnMatch1);
    sim3_direction(keys1, kfDesc1, n1, minX, minY, maxX, maxY, scaleFactors, nLevels, logScaleFactor, K4, n2, skip2, pc2in1, min2, max2, desc2, th, vnMatch2);
    int nFound = 0;
    for (int i1 = 0; i1 < n1; i1++) {
        match12[i1] = -1;
        const int idx2 = vnMatch1[i1];
        if (idx2 >= 0 && vnMatch2[idx2] == i1) { match12[i1] = idx2; nFound++; }
    }
    return nFound;
}

// SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, th, ORBdist)  (ORBmatcher.cc:1685-1793).
// kfMp[nkf]: map point id of each key-frame feature (-1 none); skip[id] = isBad() || sAlreadyFound.count(); per id: pos, min/max distance,
// descriptor.  cur_mp[ncur] in/out (-1 NULL).
int orc_search_by_projection_reloc(const KeyPoint *curKeys, const uint8_t *curDesc, int ncur, float minX, float minY, float maxX, float maxY,
                                   const float *scaleFactors, int nLevels, float logScaleFactor, const float *Tcw7, const float *Ow3,
                                   const float *K4, const KeyPoint *kfKeys, int nkf, const int32_t *kfMp, const uint8_t *skip,
                                   const float *mpPos, const float *mpMinDist, const float *mpMaxDist, const uint8_t *mpDesc, float th,
                                   int ORBdist, int checkOri, int32_t *cur_mp) {
    FrameGrid C(curKeys, curDesc, ncur, minX, minY, maxX, maxY);
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH], vIndices2;
    for (int i = 0; i < nkf; i++) {
        const int pMP = kfMp[i];
        if (pMP < 0 || skip[pMP]) continue;
        const float *x3Dw = mpPos + (size_t)pMP * 3;
        float x3Dc[3];
        se3_mul(Tcw7, x3Dw, x3Dc);
        const float u = K4[0] * x3Dc[0] / x3Dc[2] + K4[2], v = K4[1] * x3Dc[1] / x3Dc[2] + K4[3];
        if (u < minX || u > maxX) continue;
        if (v < minY || v > maxY) continue;
        const float PO[3] = {x3Dw[0] - Ow3[0], x3Dw[1] - Ow3[1], x3Dw[2] - Ow3[2]};
        const float dist3D = std::sqrt((PO[0] * PO[0] + PO[1] * PO[1]) + PO[2] * PO[2]);
        const float maxDistance = 1.2f * mpMaxDist[pMP], minDistance = 0.8f * mpMinDist[pMP];
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const int nPredictedLevel = predict_scale(mpMaxDist[pMP], dist3D, logScaleFactor, nLevels);
        const float radius = th * scaleFactors[nPredictedLevel];
        C.in_area(u, v, radius, nPredictedLevel - 1, nPredictedLevel + 1, vIndices2);
        if (vIndices2.empty()) continue;
        const uint8_t *dMP = mpDesc + (size_t)pMP * 32;
        int bestDist = 256, bestIdx2 = -1;
        for (int i2 : vIndices2) {
            if (cur_mp[i2] >= 0) continue;
            const int d = descriptor_distance(dMP, curDesc + (size_t)i2 * 32);
            if (d < bestDist) { bestDist = d; bestIdx2 = i2; }
        }
        if (bestDist <= ORBdist) {
            cur_mp[bestIdx2] = pMP;
            nmatches++;
            if (checkOri) rotHist[rot_bin(kfKeys[i].angle, curKeys[bestIdx2].angle)].push_back(bestIdx2);
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int idx : rotHist[i]) { cur_mp[idx] = -1; nmatches--; }
    }
    return nmatches;
}

// ORBmatcher::SearchForInitialization (R/lib_src/ORBmatcher.cc:581-680)
int orc_search_for_initialization(const KeyPoint *keys1, const uint8_t *desc1, int n1, const KeyPoint *keys2, const uint8_t *desc2,
                                  int n2, float minX, float minY, float maxX, float maxY, float *prevMatched, int windowSize,
                                  float nnratio, int checkOrientation, int32_t *matches12) {
    FrameGrid G2(keys2, desc2, n2, minX, minY, maxX, maxY);
    std::vector<int> cand;
    int nmatches = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<int> matchedDistance(n2, INT_MAX), matches21(n2, -1);
    for (int i1 = 0; i1 < n1; i1++) {
        const int level1 = keys1[i1].octave;
        if (level1 > 0) continue;
        G2.in_area(prevMatched[2 * i1], prevMatched[2 * i1 + 1], (float)windowSize, level1, level1, cand);
        if (cand.empty()) continue;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int i2 : cand) {
            const int dist = descriptor_distance(desc1 + (size_t)i1 * 32, desc2 + (size_t)i2 * 32);
            if (matchedDistance[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW && bestDist < (float)bestDist2 * nnratio) {
            if (matches21[bestIdx2] >= 0) { matches12[matches21[bestIdx2]] = -1; nmatches--; }
            matches12[i1] = bestIdx2;
            matches21[bestIdx2] = i1;
            matchedDistance[bestIdx2] = bestDist;
            nmatches++;
            if (checkOrientation) {
                float rot = keys1[i1].angle - keys2[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(i1);
            }
        }
    }
    if (checkOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx1 : rotHist[i])
                if (matches12[idx1] >= 0) { matches12[idx1] = -1; nmatches--; }
        }
    }
    for (int i1 = 0; i1 < n1; i1++)
        if (matches12[i1] >= 0) { prevMatched[2 * i1] = keys2[matches12[i1]].x; prevMatched[2 * i1 + 1] = keys2[matches12[i1]].y; }
    return nmatches;
}

// ORBmatcher::SearchForTriangulation, monocular branch (R/lib_src/ORBmatcher.cc:806-1013) with Pinhole::epipolarConstrain
// (Pinhole.cpp:107-129) evaluated from a caller-supplied F12 (the matrix the reference rebuilds per pair) and epipole.
int orc_search_for_triangulation(const KeyPoint *keys1, const uint8_t *desc1, int n1, const int32_t *mp1, const uint32_t *nodes1,
                                 const int32_t *off1, const uint32_t *idx1, int nn1, const KeyPoint *keys2, const uint8_t *desc2, int n2,
                                 const int32_t *mp2, const uint32_t *nodes2, const int32_t *off2, const uint32_t *idx2, int nn2,
                                 const float *scaleFactors2, const float *F12, const float *ep, int onlyStereo, int coarse,
                                 int checkOrientation, int32_t *matches12) {
    (void)n2;
    int nmatches = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    int a = 0, b = 0;
    while (a < nn1 && b < nn2) {
        if (nodes1[a] == nodes2[b]) {
            for (int p = off1[a]; p < off1[a + 1]; p++) {
                const int i1 = (int)idx1[p];
                if (mp1[i1] >= 0) continue;
                if (onlyStereo) continue;                               // bStereo1 is false for every monocular key-point
                const KeyPoint &kp1 = keys1[i1];
                int bestDist = TH_LOW, bestIdx2 = -1;
                for (int c = off2[b]; c < off2[b + 1]; c++) {
                    const int i2 = (int)idx2[c];
                    if (mp2[i2] >= 0) continue;                         // vbMatched2 is never written upstream
                    const int dist = descriptor_distance(desc1 + (size_t)i1 * 32, desc2 + (size_t)i2 * 32);
                    if (dist > TH_LOW || dist > bestDist) continue;
                    const KeyPoint &kp2 = keys2[i2];
                    const float distex = ep[0] - kp2.x, distey = ep[1] - kp2.y;
                    if (distex * distex + distey * distey < 100 * scaleFactors2[kp2.octave]) continue;
                    bool ok = coarse != 0;
                    if (!ok) {
                        const float la = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
                        const float lb = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
                        const float lc = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
                        const float num = la * kp2.x + lb * kp2.y + lc;
                        const float den = la * la + lb * lb;
                        if (den != 0) {
                            const float dsqr = num * num / den;
                            const float unc = scaleFactors2[kp2.octave] * scaleFactors2[kp2.octave];
                            ok = dsqr < 3.84 * unc;
                        }
                    }
                    if (ok) { bestIdx2 = i2; bestDist = dist; }
                }
                if (bestIdx2 >= 0) {
                    matches12[i1] = bestIdx2;
                    nmatches++;
                    if (checkOrientation) rotHist[rot_bin(kp1.angle, keys2[bestIdx2].angle)].push_back(i1);
                }
            }
            a++; b++;
        } else if (nodes1[a] < nodes2[b]) {
            while (a < nn1 && nodes1[a] < nodes2[b]) a++;               // lower_bound
        } else {
            while (b < nn2 && nodes2[b] < nodes1[a]) b++;
        }
    }
    if (checkOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int i1 : rotHist[i]) { matches12[i1] = -1; nmatches--; }
        }
    }
    return nmatches;
}

// Frame::isInFrustum(MapPoint*, viewingCosLimit), mono branch (R/lib_src/Frame.cc:558-617): fills the tracking fields of every
// point.  Rcw9 row-major = Frame::mRcw, tcw3 = mtcw, Ow3 = mOw.
void orc_is_in_frustum(const float *Rcw9, const float *tcw3, const float *Ow3, const float *K4, float minX, float minY, float maxX,
                       float maxY, float logScaleFactor, int nLevels, float viewingCosLimit, int nmp, const float *mpPos,
                       const float *mpNormal, const float *mpMinDist, const float *mpMaxDist, uint8_t *inView, float *projX,
                       float *projY, int32_t *scaleLevel, float *viewCosOut, float *trackDepth) {
    for (int i = 0; i < nmp; i++) {
        inView[i] = 0; projX[i] = -1; projY[i] = -1; scaleLevel[i] = 0; viewCosOut[i] = 0; trackDepth[i] = 0;
        const float *P = mpPos + (size_t)i * 3;
        float Pc[3];
        for (int r = 0; r < 3; r++) Pc[r] = ((Rcw9[r * 3] * P[0] + Rcw9[r * 3 + 1] * P[1]) + Rcw9[r * 3 + 2] * P[2]) + tcw3[r];
        const float Pc_dist = std::sqrt((Pc[0] * Pc[0] + Pc[1] * Pc[1]) + Pc[2] * Pc[2]);
        if (Pc[2] < 0.0f) continue;
        const float u = K4[0] * Pc[0] / Pc[2] + K4[2], v = K4[1] * Pc[1] / Pc[2] + K4[3];
        if (u < minX || u > maxX) continue;
        if (v < minY || v > maxY) continue;
        projX[i] = u; projY[i] = v;
        const float maxDistance = 1.2f * mpMaxDist[i], minDistance = 0.8f * mpMinDist[i];
        const float PO[3] = {P[0] - Ow3[0], P[1] - Ow3[1], P[2] - Ow3[2]};
        const float dist = std::sqrt((PO[0] * PO[0] + PO[1] * PO[1]) + PO[2] * PO[2]);
        if (dist < minDistance || dist > maxDistance) continue;
        const float *Pn = mpNormal + (size_t)i * 3;
        const float viewCos = ((PO[0] * Pn[0] + PO[1] * Pn[1]) + PO[2] * Pn[2]) / dist;
        if (viewCos < viewingCosLimit) continue;
        scaleLevel[i] = predict_scale(mpMaxDist[i], dist, logScaleFactor, nLevels);
        inView[i] = 1; trackDepth[i] = Pc_dist; viewCosOut[i] = viewCos;
    }
}

// Brute-force all-pairs best / second-best (the GPU formulation named by BASELINE.json config 3); ties keep the
// FIRST train index, like every matcher loop of the reference (strict <).
void orc_bruteforce_match(const uint8_t *q, int nq, const uint8_t *t, int nt, int32_t *bestIdx, int32_t *bestDist,
                          int32_t *secondDist) {
    for (int i = 0; i < nq; i++) {
        int b1 = 256, b2 = 256, bi = -1;
        for (int j = 0; j < nt; j++) {
            const int d = descriptor_distance(q + (size_t)i * 32, t + (size_t)j * 32);
            if (d < b1) { b2 = b1; b1 = d; bi = j; }
            else if (d < b2) b2 = d;
        }
        bestIdx[i] = bi; bestDist[i] = b1; secondDist[i] = b2;
    }
}

}  // extern "C"
