// ORACLE — TEST INFRASTRUCTURE ONLY.  C entry points over the CPU restatement so that tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg can drive it through ctypes.
#include <cstring>

#include "orb_oracle.h"

using orc::KeyPoint;
using orc::OrbExtractor;

extern "C" {

void *orc_orb_create(int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh) {
    return new OrbExtractor(nfeatures, scaleFactor, nlevels, iniTh, minTh);
}
void orc_orb_destroy(void *h) { delete (OrbExtractor *)h; }
void orc_orb_set_blur_variant(void *h, int v) { ((OrbExtractor *)h)->blurVariant = v; }

// returns monoIndex (>=0) or -1 (empty image) or -2 (capacity too small); *n_out = number of key-points
int orc_orb_extract(void *h, const uint8_t *img, int w, int hgt, int stride, int lap0, int lap1,
                    KeyPoint *kps, uint8_t *desc, int cap, int *n_out) {
    auto *e = (OrbExtractor *)h;
    std::vector<KeyPoint> k;
    std::vector<uint8_t> d;
    int mono = e->extract(img, w, hgt, stride, lap0, lap1, k, d);
    if (mono < 0) { *n_out = 0; return -1; }
    *n_out = (int)k.size();
    if ((int)k.size() > cap) return -2;
    if (!k.empty()) {
        std::memcpy(kps, k.data(), k.size() * sizeof(KeyPoint));
        std::memcpy(desc, d.data(), d.size());
    }
    return mono;
}

void orc_orb_tables(void *h, float *scale, float *invScale, float *sigma2, float *invSigma2,
                    int *featuresPerLevel, int *umax16) {
    auto *e = (OrbExtractor *)h;
    for (int i = 0; i < e->nlevels; i++) {
        scale[i] = e->scale[i]; invScale[i] = e->invScale[i];
        sigma2[i] = e->sigma2[i]; invSigma2[i] = e->invSigma2[i];
        featuresPerLevel[i] = e->featuresPerLevel[i];
    }
    for (int i = 0; i < 16; i++) umax16[i] = e->umax[i];
}

int orc_orb_level_size(void *h, int level, int *w, int *hgt) {
    auto *e = (OrbExtractor *)h;
    if (level < 0 || level >= e->nlevels) return -1;
    *w = e->pyr[level].w; *hgt = e->pyr[level].h;
    return 0;
}

// which: 0 = pyramid level, 1 = blurred level (empty if the level had no key-points)
int orc_orb_get_level(void *h, int level, int which, uint8_t *out) {
    auto *e = (OrbExtractor *)h;
    if (level < 0 || level >= e->nlevels) return -1;
    const orc::Image &im = which ? e->blurred[level] : e->pyr[level];
    if (im.d.empty()) return 0;
    std::memcpy(out, im.d.data(), im.d.size());
    return (int)im.d.size();
}

// which: 0 = FAST candidates before the octree (coords relative to (16,16)), 1 = selected (level coords, with angle)
int orc_orb_get_keypoints(void *h, int level, int which, KeyPoint *out, int cap) {
    auto *e = (OrbExtractor *)h;
    if (level < 0 || level >= e->nlevels) return -1;
    const auto &v = which ? e->sel[level] : e->cand[level];
    int n = (int)v.size();
    if (out && n <= cap && n) std::memcpy(out, v.data(), (size_t)n * sizeof(KeyPoint));
    return n;
}

// ---- primitives ----
int orc_cv_round(double v) { return orc::cv_round(v); }
float orc_fast_atan2(float y, float x) { return orc::fast_atan2_deg(y, x); }

void orc_resize_linear(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh) {
    orc::Image s, d;
    s.w = sw; s.h = sh; s.d.assign(src, src + (size_t)sw * sh);
    orc::resize_linear_u8(s, d, dw, dh);
    std::memcpy(dst, d.d.data(), d.d.size());
}

void orc_gaussian_blur_variant(const uint8_t *src, int w, int h, uint8_t *dst, int variant) {
    orc::Image s, d;
    s.w = w; s.h = h; s.d.assign(src, src + (size_t)w * h);
    orc::gaussian_blur_7x7_s2(s, d, variant);
    std::memcpy(dst, d.d.data(), d.d.size());
}

void orc_gaussian_blur(const uint8_t *src, int w, int h, uint8_t *dst) {
    orc::Image s, d;
    s.w = w; s.h = h; s.d.assign(src, src + (size_t)w * h);
    orc::gaussian_blur_7x7_s2(s, d);
    std::memcpy(dst, d.d.data(), d.d.size());
}

int orc_fast_score(const uint8_t *center, int stride) { return orc::fast_corner_score(center, stride); }

int orc_fast_cell(const uint8_t *img, int stride, int cols, int rows, int threshold, KeyPoint *out, int cap) {
    std::vector<KeyPoint> v;
    orc::fast_9_16_nms(img, stride, cols, rows, threshold, v);
    int n = (int)v.size();
    if (out && n <= cap && n) std::memcpy(out, v.data(), (size_t)n * sizeof(KeyPoint));
    return n;
}

int orc_octree(const KeyPoint *cand, int n, int minX, int maxX, int minY, int maxY, int N, KeyPoint *out, int cap) {
    OrbExtractor e(1000, 1.2f, 8, 20, 7);
    std::vector<KeyPoint> c(cand, cand + n);
    std::vector<KeyPoint> r = e.distribute_octree(c, minX, maxX, minY, maxY, N);
    int m = (int)r.size();
    if (out && m <= cap && m) std::memcpy(out, r.data(), (size_t)m * sizeof(KeyPoint));
    return m;
}

}  // extern "C"
