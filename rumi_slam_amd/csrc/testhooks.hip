// Host-only test hooks (include/rumi_testhooks.h): the host compilation of code the kernels share.
#include <hip/hip_runtime.h>

#include <vector>

#include "orb_geom.h"
#include "orb_math.h"
#include "orb_octree.h"
#include "rumi_orb.h"
#include <algorithm>
#include <utility>
#include <vector>

#include "rumi_testhooks.h"

using namespace rumi;

extern "C" int rumi_hook_sort_like_std(uint32_t *keys, uint16_t *ids, int32_t n) {
    if (n < 0 || (n > 0 && (!keys || !ids))) return RUMI_E_INVALID;
    std::vector<OctEntry> e((size_t)n);
    for (int i = 0; i < n; i++) e[i] = OctEntry{keys[i], ids[i], 0};
    sort_like_libstdcxx(e.data(), n);
    for (int i = 0; i < n; i++) { keys[i] = e[i].key; ids[i] = e[i].id; }
    return RUMI_OK;
}

namespace rumi { int launch_sort_hook(uint32_t *keys, uint16_t *ids, int n); }
// the same entries through the workgroup-parallel replay on the GPU (orb_octree_kernel.hip)
extern "C" int rumi_hook_sort_device(uint32_t *keys, uint16_t *ids, int32_t n) { return rumi::launch_sort_hook(keys, ids, n); }
// ... and through the real std::sort of the libstdc++ this library is built against (what the reference's compareNodes sort does)
extern "C" int rumi_hook_std_sort(uint32_t *keys, uint16_t *ids, int32_t n) {
    if (n < 0) return -1;
    std::vector<std::pair<uint32_t, uint16_t>> v(n);
    for (int i = 0; i < n; i++) v[i] = {keys[i], ids[i]};
    std::sort(v.begin(), v.end(), [](const std::pair<uint32_t, uint16_t> &a, const std::pair<uint32_t, uint16_t> &b) { return a.first < b.first; });
    for (int i = 0; i < n; i++) { keys[i] = v[i].first; ids[i] = v[i].second; }
    return 0;
}

extern "C" int rumi_hook_quadtree(const uint32_t *cand, int32_t n, int32_t minX, int32_t maxX, int32_t minY,
                                  int32_t maxY, int32_t N, int32_t *out_idx, int32_t cap, int32_t *n_out) {
    if (!n_out || n < 0 || n > 65535 || maxX <= minX || maxY <= minY) return RUMI_E_INVALID;
    std::vector<int> out;
    int m = octree_host(cand, n, minX, maxX, minY, maxY, N, out);
    if (m < 0) return RUMI_E_INVALID;
    *n_out = m;
    if (m > cap) return RUMI_E_CAPACITY;
    for (int i = 0; i < m; i++) out_idx[i] = out[i];
    return RUMI_OK;
}

extern "C" float rumi_hook_sinf(float x) { return sinf_glibc(x); }
extern "C" float rumi_hook_cosf(float x) { return cosf_glibc(x); }
extern "C" float rumi_hook_fast_atan2(float y, float x) { return fast_atan2_deg(y, x); }
extern "C" int rumi_hook_cv_round(float v) { return cv_round_f(v); }
extern "C" int rumi_hook_magic_div(int32_t idx, int32_t d) { return magic_div(idx, magic_of((unsigned)d)); }
