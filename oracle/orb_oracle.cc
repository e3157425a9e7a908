// ORACLE — TEST INFRASTRUCTURE ONLY (see orb_oracle.h for scope and parity status).
// Build: g++ -O2 -ffp-contract=off -std=c++17   (no FMA contraction: float results must match
// the reference's x86-64 SSE arithmetic and the HIP kernels, which are built the same way).
#include "orb_oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <list>

namespace orc {

static const int kPatch = 31, kHalfPatch = 15, kEdge = 19;  // R/lib_src/ORBextractor.cc:69-71

static const int8_t kPattern[256 * 4] = {
#include "orb_pattern.inc"
};

// ------------------------------------------------------------------------------------------
// OpenCV primitives, restated (upstream OpenCV 3.4 behaviour, unpinned)
// ------------------------------------------------------------------------------------------

int cv_round(double v) { return (int)std::lrint(v); }   // default FP mode = nearest-even

// cv::fastAtan2 -> hal::fastAtan32f scalar tail (atan_f32), degrees.
float fast_atan2_deg(float y, float x) {
    static const float kRad2Deg = (float)(180 / 3.1415926535897932384626433832795);
    static const float p1 = 0.9997878412794807f * kRad2Deg;
    static const float p3 = -0.3258083974640975f * kRad2Deg;
    static const float p5 = 0.1555786518463281f * kRad2Deg;
    static const float p7 = -0.04432655554792128f * kRad2Deg;
    float ax = std::fabs(x), ay = std::fabs(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// cv::resize(src, dst, Size(dw,dh), 0, 0, INTER_LINEAR) for CV_8UC1: 11-bit fixed-point taps,
// pixel-centre mapping, horizontal pass in int32 then the ((b*(S>>4))>>16 ... +2)>>2 vertical pass.
void resize_linear_u8(const Image &src, Image &dst, int dw, int dh) {
    const int sw = src.w, sh = src.h;
    dst.w = dw; dst.h = dh; dst.d.assign((size_t)dw * dh, 0);
    const double inv_x = (double)dw / sw, inv_y = (double)dh / sh;
    const double scale_x = 1. / inv_x, scale_y = 1. / inv_y;

    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> ialpha(dw * 2), ibeta(dh * 2);
    int xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)std::floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) {
            xmax = std::min(xmax, dx);
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        }
        xofs[dx] = sx;
        ialpha[dx * 2] = (short)cv_round((1.f - fx) * 2048.f);
        ialpha[dx * 2 + 1] = (short)cv_round(fx * 2048.f);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)std::floor(fy);
        fy -= sy;
        yofs[dy] = sy;
        ibeta[dy * 2] = (short)cv_round((1.f - fy) * 2048.f);
        ibeta[dy * 2 + 1] = (short)cv_round(fy * 2048.f);
    }
    auto clip = [](int v, int lo, int hi) { return v >= lo ? (v < hi ? v : hi - 1) : lo; };
    std::vector<int> r0(dw), r1(dw);
    int have0 = -1, have1 = -1;
    auto hrow = [&](int sy, std::vector<int> &out) {
        const uint8_t *S = src.row(sy);
        int dx = 0;
        for (; dx < xmax; dx++) {
            int sx = xofs[dx];
            out[dx] = S[sx] * ialpha[dx * 2] + S[sx + 1] * ialpha[dx * 2 + 1];
        }
        for (; dx < dw; dx++) out[dx] = S[xofs[dx]] * 2048;
    };
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = clip(yofs[dy], 0, sh), sy1 = clip(yofs[dy] + 1, 0, sh);
        if (sy0 == have1) { r0.swap(r1); std::swap(have0, have1); }
        if (sy0 != have0) { hrow(sy0, r0); have0 = sy0; }
        if (sy1 == have0) { r1 = r0; have1 = sy1; }
        else if (sy1 != have1) { hrow(sy1, r1); have1 = sy1; }
        const int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
        uint8_t *D = dst.row(dy);
        for (int x = 0; x < dw; x++)
            D[x] = (uint8_t)((((b0 * (r0[x] >> 4)) >> 16) + ((b1 * (r1[x] >> 4)) >> 16) + 2) >> 2);
    }
}

static inline int reflect101(int i, int n) {   // BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

// cv::GaussianBlur(Size(7,7), 2, 2, BORDER_REFLECT_101) on CV_8UC1.  The reference pins "OpenCV 3.4" only (R/CMakeLists.txt:35) and two
// implementations exist in that series; `variant` selects which one is restated (parity unpinned either way: the reference holds no vector):
//   0  the bit-exact fixed-point path (3.4.2 and later, 4.x): kernel quantised to 8 fractional bits with the sum corrected to 256,
//      taps {18,34,48,56,48,34,18}/256, 8.8 row pass into u16, 16.16 column pass, one rounding at the end: (v + 32768) >> 16;
//   1  the sepFilter2D path of 3.4.0 / 3.4.1: getGaussianKernel(7, 2, CV_32F) = {.070159,.131075,.190713,.216106,...} converted tap by tap with
//      convertTo(CV_32S, 256) -> {18,34,49,55,49,34,18} (sum 257), integer row and column passes, FixedPtCastEx: saturate((v + 32768) >> 16).
void gaussian_blur_7x7_s2(const Image &src, Image &dst, int variant) {
    static const int kFixed[7] = {18, 34, 48, 56, 48, 34, 18}, kSep[7] = {18, 34, 49, 55, 49, 34, 18};
    const int *k = variant ? kSep : kFixed;
    const int w = src.w, h = src.h;
    dst.w = w; dst.h = h; dst.d.assign((size_t)w * h, 0);
    std::vector<uint32_t> tmp((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t *S = src.row(y);
        uint32_t *T = tmp.data() + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            int acc = 0;
            if (x >= 3 && x + 3 < w)
                for (int i = 0; i < 7; i++) acc += k[i] * S[x + i - 3];
            else
                for (int i = 0; i < 7; i++) acc += k[i] * S[reflect101(x + i - 3, w)];
            T[x] = (uint32_t)acc;
        }
    }
    for (int y = 0; y < h; y++) {
        const uint32_t *R[7];
        for (int j = 0; j < 7; j++) R[j] = tmp.data() + (size_t)reflect101(y + j - 3, h) * w;
        uint8_t *D = dst.row(y);
        for (int x = 0; x < w; x++) {
            uint32_t acc = 0;
            for (int j = 0; j < 7; j++) acc += (uint32_t)k[j] * R[j][x];
            const uint32_t v = (acc + 32768u) >> 16;
            D[x] = (uint8_t)(v > 255u ? 255u : v);
        }
    }
}

// FAST 9/16 circle, cv order.
static const int kCx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int kCy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

// cornerScore<16> with the starting threshold folded out: returns max(A,B)-1 where
// A = max over the 16 arcs of 9 of min(v - I_k), B = the same on (I_k - v).  A pixel is a FAST
// corner at threshold T  <=>  this value >= T, and for corners it equals cv's score.
int fast_corner_score(const uint8_t *p, int stride) {
    int d[25];
    const int v = p[0];
    for (int k = 0; k < 16; k++) d[k] = v - p[kCy[k] * stride + kCx[k]];
    for (int k = 16; k < 25; k++) d[k] = d[k - 16];
    int A = -256, B = -256;
    for (int k = 0; k < 16; k++) {
        int mn = d[k], mx = d[k];
        for (int j = 1; j < 9; j++) { mn = std::min(mn, d[k + j]); mx = std::max(mx, d[k + j]); }
        A = std::max(A, mn);
        B = std::max(B, -mx);
    }
    return std::max(A, B) - 1;
}

// cv::FAST(img[rows x cols], kps, threshold, nonmaxSuppression=true), TYPE_9_16.
// Rows/cols 0..2 and the last 3 are never tested; scores of non-corners and of everything outside
// the tested interior are 0; a corner survives iff its score is strictly greater than its 8
// neighbours'; output row-major as (x, y, size 7, angle -1, response = score).
void fast_9_16_nms(const uint8_t *img, int stride, int cols, int rows, int threshold,
                   std::vector<KeyPoint> &out) {
    if (cols < 7 || rows < 7) return;
    threshold = std::min(std::max(threshold, 0), 255);
    std::vector<uint8_t> sc((size_t)cols * rows, 0);
    for (int y = 3; y < rows - 3; y++) {
        const uint8_t *p = img + (size_t)y * stride;
        uint8_t *s = sc.data() + (size_t)y * cols;
        for (int x = 3; x < cols - 3; x++) {
            // corner test first (as cv::FAST does), score only for corners:
            // a 9-arc must contain circle pixel 0 or 8 and pixel 4 or 12 -> cheap reject ...
            const int v = p[x], lo = v - threshold, hi = v + threshold;
            const int a = p[x + 3 * stride], b = p[x - 3 * stride];
            if (!((a > hi) | (a < lo) | (b > hi) | (b < lo))) continue;
            const int c = p[x + 3], e = p[x - 3];
            if (!((c > hi) | (c < lo) | (e > hi) | (e < lo))) continue;
            // ... then the exact test: 9 contiguous circle pixels all brighter or all darker.
            unsigned mb = 0, md = 0;
            for (int k = 0; k < 16; k++) {
                const int q = p[x + kCy[k] * stride + kCx[k]];
                mb |= (unsigned)(q > hi) << k;
                md |= (unsigned)(q < lo) << k;
            }
            auto arc9 = [](unsigned m) {
                m |= m << 16;                     // unroll the circle
                unsigned r = m & (m >> 1);        // runs of 2
                r &= r >> 2;                      // runs of 4
                r &= r >> 4;                      // runs of 8
                r &= m >> 8;                      // runs of 9
                return (r & 0xffffu) != 0;
            };
            if (!arc9(mb) && !arc9(md)) continue;
            int score = fast_corner_score(p + x, stride);
            if (score >= threshold) s[x] = (uint8_t)score;
        }
    }
    for (int y = 3; y < rows - 3; y++) {
        const uint8_t *s = sc.data() + (size_t)y * cols;
        for (int x = 3; x < cols - 3; x++) {
            int v = s[x];
            if (!v) continue;
            if (v > s[x - 1] && v > s[x + 1] && v > s[x - cols - 1] && v > s[x - cols] && v > s[x - cols + 1] &&
                v > s[x + cols - 1] && v > s[x + cols] && v > s[x + cols + 1])
                out.push_back(KeyPoint{(float)x, (float)y, 7.f, -1.f, (float)v, 0, -1});
        }
    }
}

// ------------------------------------------------------------------------------------------
// ORBextractor
// ------------------------------------------------------------------------------------------

// R/lib_src/ORBextractor.cc:405-461
OrbExtractor::OrbExtractor(int nf, float sf, int nl, int ini, int mn)
    : nfeatures(nf), nlevels(nl), iniTh(ini), minTh(mn), scaleFactor(sf) {
    scale.resize(nlevels); sigma2.resize(nlevels); invScale.resize(nlevels); invSigma2.resize(nlevels);
    scale[0] = 1.f; sigma2[0] = 1.f;
    for (int i = 1; i < nlevels; i++) {
        scale[i] = (float)(scale[i - 1] * scaleFactor);   // float * double -> double -> float
        sigma2[i] = scale[i] * scale[i];
    }
    for (int i = 0; i < nlevels; i++) { invScale[i] = 1.0f / scale[i]; invSigma2[i] = 1.0f / sigma2[i]; }

    featuresPerLevel.resize(nlevels);
    float factor = (float)(1.0f / scaleFactor);
    float nDesired = (float)(nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels)));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) {
        featuresPerLevel[l] = cv_round(nDesired);
        sum += featuresPerLevel[l];
        nDesired *= factor;
    }
    featuresPerLevel[nlevels - 1] = std::max(nfeatures - sum, 0);

    umax.assign(kHalfPatch + 1, 0);
    int vmax = (int)std::floor(kHalfPatch * std::sqrt(2.f) / 2 + 1);
    int vmin = (int)std::ceil(kHalfPatch * std::sqrt(2.f) / 2);
    const double hp2 = kHalfPatch * kHalfPatch;
    for (int v = 0; v <= vmax; ++v) umax[v] = cv_round(std::sqrt(hp2 - v * v));
    for (int v = kHalfPatch, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
    pyr.resize(nlevels); blurred.resize(nlevels); cand.resize(nlevels); sel.resize(nlevels);
}

// R/lib_src/ORBextractor.cc:1093-1112.  The 19-px REFLECT_101 border the reference adds around each
// level is never read on the mono path, so levels are stored border-less here.
void OrbExtractor::compute_pyramid(const uint8_t *img, int w, int h, int stride) {
    for (int l = 0; l < nlevels; l++) {
        float s = invScale[l];
        int lw = cv_round((float)w * s), lh = cv_round((float)h * s);
        if (l == 0) {
            pyr[0].w = w; pyr[0].h = h; pyr[0].d.resize((size_t)w * h);
            for (int y = 0; y < h; y++) std::memcpy(pyr[0].row(y), img + (size_t)y * stride, w);
        } else {
            resize_linear_u8(pyr[l - 1], pyr[l], lw, lh);
        }
    }
}

// R/lib_src/ORBextractor.cc:726-808 — the per-cell cv::FAST loop of one level.
void OrbExtractor::detect_candidates(int level, std::vector<KeyPoint> &out) const {
    const Image &im = pyr[level];
    const int minBX = kEdge - 3, minBY = minBX;
    const int maxBX = im.w - kEdge + 3, maxBY = im.h - kEdge + 3;
    const float W = 35;
    const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
    const int nCols = (int)(width / W), nRows = (int)(height / W);
    if (nCols <= 0 || nRows <= 0) return;
    const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
    std::vector<KeyPoint> cell;
    for (int i = 0; i < nRows; i++) {
        const float iniY = (float)(minBY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBY - 3) continue;
        if (maxY > maxBY) maxY = (float)maxBY;
        for (int j = 0; j < nCols; j++) {
            const float iniX = (float)(minBX + j * wCell);
            float maxX = iniX + wCell + 6;
            if (iniX >= maxBX - 6) continue;
            if (maxX > maxBX) maxX = (float)maxBX;
            const int x0 = (int)iniX, x1 = (int)maxX, y0 = (int)iniY, y1 = (int)maxY;
            cell.clear();
            fast_9_16_nms(im.row(y0) + x0, im.w, x1 - x0, y1 - y0, iniTh, cell);
            if (cell.empty()) fast_9_16_nms(im.row(y0) + x0, im.w, x1 - x0, y1 - y0, minTh, cell);
            for (KeyPoint kp : cell) {
                kp.x += j * wCell;
                kp.y += i * hCell;
                out.push_back(kp);
            }
        }
    }
}

namespace {
struct Node {                 // ExtractorNode: rectangle [x0,x1) x [y0,y1) + its keys in parent order
    int x0, x1, y0, y1;
    std::vector<int> keys;    // indices into the candidate array
    bool noMore = false;
};
using NodeList = std::list<Node>;
using SizedNode = std::pair<int, NodeList::iterator>;

// ExtractorNode::DivideNode, R/lib_src/ORBextractor.cc:471-522
void divide(const Node &p, const std::vector<KeyPoint> &kp, Node c[4]) {
    const int hx = (int)std::ceil((float)(p.x1 - p.x0) / 2), hy = (int)std::ceil((float)(p.y1 - p.y0) / 2);
    c[0] = Node{p.x0, p.x0 + hx, p.y0, p.y0 + hy, {}, false};
    c[1] = Node{p.x0 + hx, p.x1, p.y0, p.y0 + hy, {}, false};
    c[2] = Node{p.x0, p.x0 + hx, p.y0 + hy, p.y1, {}, false};
    c[3] = Node{p.x0 + hx, p.x1, p.y0 + hy, p.y1, {}, false};
    const float sx = (float)(p.x0 + hx), sy = (float)(p.y0 + hy);
    for (int k : p.keys) {
        int q = (kp[k].x < sx ? 0 : 1) + (kp[k].y < sy ? 0 : 2);
        c[q].keys.push_back(k);
    }
    for (int q = 0; q < 4; q++) c[q].noMore = c[q].keys.size() == 1;
}

// compareNodes, R/lib_src/ORBextractor.cc:524-536
bool node_less(const SizedNode &a, const SizedNode &b) {
    if (a.first != b.first) return a.first < b.first;
    return a.second->x0 < b.second->x0;
}
}  // namespace

// R/lib_src/ORBextractor.cc:538-724.  Same container choreography as the reference (std::list with
// push_front, std::sort on (size, UL.x)) because the output ORDER and the tie order of the sort are
// part of the result.
std::vector<KeyPoint> OrbExtractor::distribute_octree(const std::vector<KeyPoint> &kp, int minX, int maxX,
                                                      int minY, int maxY, int N) const {
    std::vector<KeyPoint> result;
    const int nIni = (int)std::round((float)(maxX - minX) / (maxY - minY));
    if (nIni <= 0) return result;
    const float hX = (float)(maxX - minX) / nIni;
    NodeList nodes;
    std::vector<NodeList::iterator> roots(nIni);
    for (int i = 0; i < nIni; i++) {
        nodes.push_back(Node{(int)(hX * (float)i), (int)(hX * (float)(i + 1)), 0, maxY - minY, {}, false});
        roots[i] = std::prev(nodes.end());
    }
    for (int k = 0; k < (int)kp.size(); k++) roots[(int)(kp[k].x / hX)]->keys.push_back(k);
    for (auto it = nodes.begin(); it != nodes.end();) {
        if (it->keys.size() == 1) { it->noMore = true; ++it; }
        else if (it->keys.empty()) it = nodes.erase(it);
        else ++it;
    }

    std::vector<SizedNode> open;
    auto push_children = [&](Node c[4], int *nExpand) {
        for (int q = 0; q < 4; q++) {
            if (c[q].keys.empty()) continue;
            nodes.push_front(std::move(c[q]));
            if (nodes.front().keys.size() > 1) {
                if (nExpand) ++*nExpand;
                open.emplace_back((int)nodes.front().keys.size(), nodes.begin());
            }
        }
    };

    bool done = false;
    while (!done) {
        int prevSize = (int)nodes.size(), nExpand = 0;
        open.clear();
        for (auto it = nodes.begin(); it != nodes.end();) {
            if (it->noMore) { ++it; continue; }
            Node c[4];
            divide(*it, kp, c);
            push_children(c, &nExpand);
            it = nodes.erase(it);
        }
        if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) {
            done = true;
        } else if ((int)nodes.size() + nExpand * 3 > N) {
            while (!done) {
                prevSize = (int)nodes.size();
                std::vector<SizedNode> prev = open;
                open.clear();
                std::sort(prev.begin(), prev.end(), node_less);
                for (int j = (int)prev.size() - 1; j >= 0; j--) {
                    Node c[4];
                    divide(*prev[j].second, kp, c);
                    push_children(c, nullptr);
                    nodes.erase(prev[j].second);
                    if ((int)nodes.size() >= N) break;
                }
                if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) done = true;
            }
        }
    }
    result.reserve(nodes.size());
    for (const Node &n : nodes) {
        int best = n.keys[0];
        for (size_t k = 1; k < n.keys.size(); k++)
            if (kp[n.keys[k]].response > kp[best].response) best = n.keys[k];
        result.push_back(kp[best]);
    }
    return result;
}

// IC_Angle, R/lib_src/ORBextractor.cc:73-97
float OrbExtractor::ic_angle(const Image &im, float x, float y) const {
    int m01 = 0, m10 = 0;
    const int step = im.w;
    const uint8_t *c = im.row(cv_round(y)) + cv_round(x);
    for (int u = -kHalfPatch; u <= kHalfPatch; ++u) m10 += u * c[u];
    for (int v = 1; v <= kHalfPatch; ++v) {
        int vs = 0, d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int p = c[u + v * step], m = c[u - v * step];
            vs += p - m;
            m10 += u * (p + m);
        }
        m01 += v * vs;
    }
    return fast_atan2_deg((float)m01, (float)m10);
}

// computeOrbDescriptor, R/lib_src/ORBextractor.cc:99-143
void OrbExtractor::orb_descriptor(const Image &im, const KeyPoint &kp, uint8_t *desc) const {
    static const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float angle = (float)kp.angle * factorPI;
    // The reference writes `(float)cos(angle)` with a float argument under `using namespace std`, which
    // overload resolution binds to std::cos(float) = libm cosf (single precision), not double cos.
    float a = std::cos(angle), b = std::sin(angle);
    static_assert(sizeof(decltype(std::cos(angle))) == sizeof(float), "float overload");
    const uint8_t *c = im.row(cv_round(kp.y)) + cv_round(kp.x);
    const int step = im.w;
    auto val = [&](int idx) -> int {
        const int px = kPattern[idx * 2], py = kPattern[idx * 2 + 1];
        return c[cv_round(px * b + py * a) * step + cv_round(px * a - py * b)];
    };
    for (int i = 0; i < 32; i++) {
        int byte = 0;
        for (int k = 0; k < 8; k++) {
            int t0 = val(i * 16 + 2 * k), t1 = val(i * 16 + 2 * k + 1);
            byte |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)byte;
    }
}

// ORBextractor::operator(), R/lib_src/ORBextractor.cc:1014-1091 (+ ComputeKeyPointsOctTree :726-831)
int OrbExtractor::extract(const uint8_t *img, int w, int h, int stride, int lap0, int lap1,
                          std::vector<KeyPoint> &kps, std::vector<uint8_t> &desc) {
    if (!img || w <= 0 || h <= 0) return -1;
    compute_pyramid(img, w, h, stride);
    int total = 0;
    for (int l = 0; l < nlevels; l++) {
        const Image &im = pyr[l];
        const int minBX = kEdge - 3, minBY = minBX, maxBX = im.w - kEdge + 3, maxBY = im.h - kEdge + 3;
        cand[l].clear();
        detect_candidates(l, cand[l]);
        sel[l] = distribute_octree(cand[l], minBX, maxBX, minBY, maxBY, featuresPerLevel[l]);
        const int scaledPatch = (int)(kPatch * scale[l]);
        for (KeyPoint &k : sel[l]) {
            k.x += minBX; k.y += minBY; k.octave = l; k.size = (float)scaledPatch;
        }
        total += (int)sel[l].size();
    }
    for (int l = 0; l < nlevels; l++)
        for (KeyPoint &k : sel[l]) k.angle = ic_angle(pyr[l], k.x, k.y);

    kps.assign(total, KeyPoint{});
    desc.assign((size_t)total * 32, 0);
    int mono = 0, stereo = total - 1;
    for (int l = 0; l < nlevels; l++) {
        if (sel[l].empty()) { blurred[l] = Image{}; continue; }
        gaussian_blur_7x7_s2(pyr[l], blurred[l], blurVariant);
        const float s = scale[l];
        for (const KeyPoint &k0 : sel[l]) {
            uint8_t d[32];
            orb_descriptor(blurred[l], k0, d);
            KeyPoint k = k0;
            if (l != 0) { k.x *= s; k.y *= s; }
            int slot = (k.x >= lap0 && k.x <= lap1) ? stereo-- : mono++;
            kps[slot] = k;
            std::memcpy(&desc[(size_t)slot * 32], d, 32);
        }
    }
    return mono;
}

}  // namespace orc
