// Host-side geometry and constant tables of the ORB extractor (no device code here).
// Reference: R/lib_src/ORBextractor.cc:405-461 (constructor), :729-763 (cell grid), :1093-1112
// (level sizes); OpenCV 3.4 resize coefficient rule.
#pragma once
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <vector>

namespace rumi {

constexpr int kPatchSize = 31;
constexpr int kHalfPatch = 15;
constexpr int kEdge = 19;          // EDGE_THRESHOLD
constexpr int kBorder = kEdge - 3; // minBorderX/Y = 16
constexpr int kMaxLevels = 16;
constexpr int kCellTileMax = 96;   // largest FAST sub-image side the cell kernel stages in LDS
// No border is materialised around a level: the only reader outside a level is the 7x7 blur, which mirrors (BORDER_REFLECT_101) at the
// edges itself; the 19-px border of mvImagePyramid is synthesised on export (rumi_orb_pyramid_level).

inline int cv_round_host(double v) { return (int)std::lrint(v); }

// idx / d without an integer divide, for an index into a tile of at most 98 rows of d <= 98 elements (idx < 99 d):
// M = floor(2^20 / d) + 1, q = (idx * M) >> 20.  With M d = 2^20 + f, 0 < f <= d, and idx = q d + r:
// idx M = q 2^20 + (q f + r M) with q f + r M < 2^20 because q <= 98 and f <= d <= 98; idx M < 99 * 2^20 * 1.0001 < 2^32.
// Checked exhaustively by tests/test_hostcode_cpu.py through rumi_hook_magic_div.
#if defined(__HIPCC__)
#define RUMI_GEOM_HD __host__ __device__ inline
#else
#define RUMI_GEOM_HD inline
#endif
RUMI_GEOM_HD unsigned magic_of(unsigned d) { return (1u << 20) / d + 1u; }
RUMI_GEOM_HD int magic_div(int idx, unsigned M) { return (int)(((unsigned)idx * M) >> 20); }
// plain 32-bit products: v_mul_lo_u32 issues at full rate on gfx950 (profiles/r01_valu_issue_rates.txt), and the 24-bit intrinsics cost an
// extra mask per operand where the compiler cannot prove the range (measured: +0.8 % instructions in k_fast_cells)
RUMI_GEOM_HD int mul24(int a, int b) { return a * b; }

struct LevelGeom {
    int w, h, pitch;            // level size, row pitch in bytes (64-B aligned)
    long long off;              // byte offset of the level's pixel (0,0) inside one frame's arena
    int nCols, nRows, wCell, hCell;
    int cellBase, nCells;       // first cell id of this level in the frame's cell list
    int maxBX, maxBY;           // w-16, h-16
    int nfeat;                  // mnFeaturesPerLevel
    float scale;                // mvScaleFactor
    int candBase;               // first slot of this level in the frame's worst-case candidate arena
    int candCap;
    int coefOff;                // offset (in int16 units) of this level's resize tables
};

struct OrbTables {
    int nlevels = 0;
    std::vector<float> scale, invScale, sigma2, invSigma2;
    std::vector<int> featuresPerLevel, umax;
};

// ORBextractor::ORBextractor, R/lib_src/ORBextractor.cc:405-461
inline OrbTables make_tables(int nfeatures, float scaleFactorArg, int nlevels) {
    OrbTables t;
    t.nlevels = nlevels;
    const double scaleFactor = scaleFactorArg;   // the member is a double initialised from the float arg
    t.scale.assign(nlevels, 1.f); t.sigma2.assign(nlevels, 1.f);
    t.invScale.resize(nlevels); t.invSigma2.resize(nlevels);
    for (int i = 1; i < nlevels; i++) {
        t.scale[i] = (float)(t.scale[i - 1] * scaleFactor);
        t.sigma2[i] = t.scale[i] * t.scale[i];
    }
    for (int i = 0; i < nlevels; i++) { t.invScale[i] = 1.0f / t.scale[i]; t.invSigma2[i] = 1.0f / t.sigma2[i]; }
    t.featuresPerLevel.assign(nlevels, 0);
    float factor = (float)(1.0f / scaleFactor);
    float want = (float)(nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels)));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) {
        t.featuresPerLevel[l] = cv_round_host(want);
        sum += t.featuresPerLevel[l];
        want *= factor;
    }
    t.featuresPerLevel[nlevels - 1] = std::max(nfeatures - sum, 0);
    t.umax.assign(kHalfPatch + 1, 0);
    int vmax = (int)std::floor(kHalfPatch * std::sqrt(2.f) / 2 + 1);
    int vmin = (int)std::ceil(kHalfPatch * std::sqrt(2.f) / 2);
    const double hp2 = kHalfPatch * kHalfPatch;
    for (int v = 0; v <= vmax; ++v) t.umax[v] = cv_round_host(std::sqrt(hp2 - v * v));
    for (int v = kHalfPatch, v0 = 0; v >= vmin; --v) {
        while (t.umax[v0] == t.umax[v0 + 1]) ++v0;
        t.umax[v] = v0;
        ++v0;
    }
    return t;
}

// Level sizes (ComputePyramid :1095-1096) and FAST cell grids (:729-746) for a w x h frame.
// Returns false when a level is too small for the FAST cell loop or a cell exceeds the LDS tile.
inline bool make_geometry(const OrbTables &t, int w, int h, std::vector<LevelGeom> &g,
                          long long *arenaBytes, int *totalCells, int *totalCand, int *maxCellCand) {
    g.assign(t.nlevels, LevelGeom{});
    long long off = 0;
    int cells = 0, cand = 0, cellCand = 1;
    for (int l = 0; l < t.nlevels; l++) {
        LevelGeom &L = g[l];
        L.w = cv_round_host((float)w * t.invScale[l]);
        L.h = cv_round_host((float)h * t.invScale[l]);
        L.pitch = (L.w + 63) & ~63;
        L.off = off;
        off += (long long)L.pitch * L.h + 64;             // + slack: the resize reads 8-byte windows that may end a few bytes past a row
        L.maxBX = L.w - kBorder; L.maxBY = L.h - kBorder;
        const float width = (float)(L.maxBX - kBorder), height = (float)(L.maxBY - kBorder);
        L.nCols = (int)(width / 35.f); L.nRows = (int)(height / 35.f);
        if (L.nCols <= 0 || L.nRows <= 0) return false;      // the reference divides by zero here
        L.wCell = (int)std::ceil(width / L.nCols); L.hCell = (int)std::ceil(height / L.nRows);
        if (L.wCell + 6 > kCellTileMax || L.hCell + 6 > kCellTileMax) return false;
        L.cellBase = cells; L.nCells = L.nCols * L.nRows; cells += L.nCells;
        L.nfeat = t.featuresPerLevel[l];
        L.scale = t.scale[l];
        // NMS keeps at most one of any 2x2 block: worst case per level / per cell
        const int dw = L.maxBX - kBorder - 6, dh = L.maxBY - kBorder - 6;   // detection region
        L.candBase = cand;
        L.candCap = std::min(65535, std::max(1, ((dw + 1) / 2) * ((dh + 1) / 2)));   // u16 key ids in the quadtree
        cand += L.candCap;
        cellCand = std::max(cellCand, ((L.wCell + 1) / 2) * ((L.hCell + 1) / 2));
        L.coefOff = 0;
    }
    *arenaBytes = (off + 255) & ~255LL;
    *totalCells = cells; *totalCand = cand; *maxCellCand = cellCand;
    return true;
}

// cv::resize INTER_LINEAR 8UC1 coefficient tables for one axis: ofs[d] and the two 11-bit taps.
// Horizontal axis (clampX = true): the source index is clamped and the tap zeroed as cv does, and
// `maxOut` receives xmax (first destination index whose second tap would fall outside the source:
// from there on the row pass emits S[sx]*2048).  Vertical axis (clampX = false): cv keeps the raw
// index and fractional tap and clips the two ROW indices when it fetches them, so we do the same.
inline void make_resize_axis(int srcN, int dstN, bool clampX, std::vector<int16_t> &ofs,
                             std::vector<int16_t> &taps, int *maxOut) {
    const double inv = (double)dstN / srcN, scale = 1. / inv;
    ofs.resize(dstN); taps.resize(dstN * 2);
    int dmax = dstN;
    for (int d = 0; d < dstN; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(f);
        f -= s;
        if (clampX) {
            if (s < 0) { f = 0; s = 0; }
            if (s + 1 >= srcN) {
                dmax = std::min(dmax, d);
                if (s >= srcN - 1) { f = 0; s = srcN - 1; }
            }
        }
        ofs[d] = (int16_t)s;
        taps[d * 2] = (int16_t)cv_round_host((1.f - f) * 2048.f);
        taps[d * 2 + 1] = (int16_t)cv_round_host(f * 2048.f);
    }
    *maxOut = dmax;
}

}  // namespace rumi
