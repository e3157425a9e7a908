// Host side of the C ABI in include/rumi_orb.h: handle, HBM arenas, stage scheduling.
// Reference behaviour: ORBextractor::operator() R/lib_src/ORBextractor.cc:1014-1091.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "orb_device.h"
#include "orb_octree.h"
#include "rumi_common.h"

using namespace rumi;

namespace rumi {
thread_local std::string g_lastError;
void set_error(const char *fmt, const char *a, const char *b, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, fmt, a, b, line);
    g_lastError = buf;
}
}  // namespace rumi

extern "C" const char *rumi_last_error(void) { return g_lastError.c_str(); }

extern "C" int rumi_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

namespace {
constexpr int kChunk = 256;  // frames per pass through the candidate / quadtree scratch arenas
static int scratch_frames(int maxBatch) { return (std::min(kChunk, maxBatch) + 11) / 12 * 12; }

template <class T> int dev_alloc(T **p, size_t n) {
    *p = nullptr;
    HIP_TRY(hipMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(T)));
    return RUMI_OK;
}
template <class T> int pin_alloc(T **p, size_t n) {
    *p = nullptr;
    HIP_TRY(hipHostMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(T), hipHostMallocDefault));
    return RUMI_OK;
}
}  // namespace

struct RumiOrb {
    RumiOrbConfig cfg{};
    OrbTables tab;
    int device = 0;
    int hostThreads = 1;
    // capacities (from max_width x max_height)
    long long capArena = 0;
    int capCells = 0, capCand = 0, capCellCand = 0, capSel = 0, capCoef = 0;
    // geometry of the current image size
    int gw = 0, gh = 0;
    DevParams hP{};
    DevParams *dP = nullptr;
    int16_t *dCoef = nullptr;
    RowTap *dRowTab = nullptr; int capRowTab = 0;   // per level and output row: source rows and vertical taps of the resize
    PyrTile *dPyrTiles[2] = {nullptr, nullptr}; int nPyrTiles[2] = {0, 0}, pyrBuf[2] = {0, 0}, pyrTab[2] = {0, 0};   // the one-launch pyramid (k_pyramid_tiles), [0] small tiles for calls of a few frames, [1] large tiles for batches: 0 tiles = not available for this geometry
    // HBM arenas (sized for max_batch frames unless noted)
    uint8_t *dIn = nullptr;          // staging for the single-frame host API (1 frame, rows padded to a multiple of 4 bytes)
    uint8_t *dL0 = nullptr;          // staging for device frames whose base / pitch / frame stride is not 4-byte aligned (allocated on first use)
    uint8_t *hIn = nullptr, *hOut1 = nullptr, *dOut1 = nullptr;   // pinned image / pinned + device [counts | kp | desc] block of that API
    size_t out1Bytes = 0;             // > 0 while rumi_orb_extract wants the block copied back before the call's one synchronisation
    uint8_t *dhOut1 = nullptr;        // the pinned block as the device addresses it: a one-frame call's kernels write counts, key-points and descriptors
    int32_t *dhErr = nullptr;         // straight into host memory (and k_assemble the final error word): no copy back, the call ends with its last kernel
    bool zeroCopyOut = false;         // set by rumi_orb_extract around its call
    uint8_t *dhIn = nullptr;          // the pinned image as the device addresses it
    bool hostImagePending = false;    // rumi_orb_extract: the frame of this call still sits in hIn (w x hgt, pitch wp): extract_async_impl either lets the
                                      // one-launch pyramid read it over PCIe (and keep a copy as the arena's level 0) or copies it to dIn first
    uint8_t *dPyr = nullptr, *dBlur = nullptr;
    uint32_t *dCellBuf = nullptr;    // kChunk frames
    int32_t *dCellCnt = nullptr;
    uint32_t *dCand = nullptr;       // kChunk frames x capCand
    int32_t *dLevelStart = nullptr;
    uint32_t *dSelPacked = nullptr, *dSelMeta = nullptr;   // kChunk frames x capSel
    int32_t *dSelCount = nullptr;
    RumiKeyPoint *dKp = nullptr;     // outputs of the single-frame host API
    uint8_t *dDesc = nullptr;
    int32_t *dCounts = nullptr;
    uint16_t *dOwner = nullptr;      // kChunk frames x capCand: quadtree node id of every candidate
    uint32_t *dSelLevel = nullptr;   // kChunk frames x nlevels x selLevelCap: quadtree output per level
    int32_t *dSelLevelCnt = nullptr;
    int32_t *dErr = nullptr;         // device error word (bit 0/1/2/3: roots, node pool, level cap, selection cap; bit 4: FAST candidate capacity);
                                     // sticky over the asynchronous calls since the last rumi_orb_sync
    // host-resident batches (rumi_orb_extract_batch_host): device landing arena, pinned staging slots, copy stream; `feed`, when set, makes the
    // frames [0, upto) of the running call resident and lets stream s wait for them
    uint8_t *dHostIn = nullptr; size_t dHostInBytes = 0;
    static constexpr int kFeedSlots = 4, kFeedFrames = 64;
    uint8_t *hFeed[kFeedSlots] = {nullptr}; size_t hFeedBytes = 0;
    hipEvent_t evFeed[kFeedSlots] = {nullptr};
    hipStream_t copyStream = nullptr, copyStream2 = nullptr;     // host -> device transfers of rumi_orb_extract_batch_host, groups alternating (two DMA engines)
    std::function<int(int, hipStream_t)> feed;
    bool pending = false;            // an asynchronous call has been enqueued and not yet waited for
    hipStream_t pendingStream = nullptr;
    int selLevelCap = 0;
    size_t octLds = 0;
    // pinned host words
    int32_t *hErr = nullptr;
    // host copies fetched lazily by the stage taps
    std::vector<uint32_t> tapCand, tapSelPacked, tapSelMeta;
    std::vector<int32_t> tapLevelStart, tapSelCount;
    bool tapValid = false;
    // last-call bookkeeping for the stage taps
    ImgSrc lastSrc{};
    int lastFrames = 0, lastChunkBase = 0, lastChunkFrames = 0, lastChunkSlot = 0;   // frames / scratch slot the stage taps can read
    RumiKeyPoint *lastKp = nullptr;  // device pointer the last call wrote key-points to
    long long lastKpStride = 0;      // bytes between the key-points of consecutive frames there
    int32_t *lastCounts = nullptr;
    int lastOutCap = 0;
    bool profiling = false;
    float stageMs[8] = {0};
    // host entries with PINNED host destinations: every sub-chunk's output rows [frame0, frame0 + n) follow its kernels to the host on the sub-chunk's own
    // stream, under the kernels of the sub-chunks behind it (one entry for the record layout, three for key-points / descriptors / counts)
    struct Mirror { uint8_t *host; const uint8_t *dev; long long row; } mirror[3] = {{nullptr, nullptr, 0}, {nullptr, nullptr, 0}, {nullptr, nullptr, 0}};
    hipEvent_t ev[8] = {nullptr};
    // the blur only depends on the pyramid: it runs on a side stream next to FAST / quadtree and joins before rBRIEF
    hipStream_t sideStream = nullptr;
    // a chunk's frames are split over the caller's stream and these, see Stage B
    static constexpr int kMaxParts = 8;
    hipStream_t partStream[kMaxParts - 1] = {nullptr}, partSide[kMaxParts - 1] = {nullptr};   // partSide: the part's blur when the pyramid is split too
    // rumi_orb_set_resident_queue: the frames of a call do not depend on work pending on the caller's stream, so sub-chunk 0 gets a stream of
    // its own too (with its blur stream) and no sub-chunk waits for the caller's stream: back-to-back calls then overlap like the sub-chunks of
    // one large call, only the caller's stream waits for each call's results
    bool residentQueue = false;
    int residentSlots = 4;                    // slots of the resident queue (2 .. kMaxParts)
    int scratchFrames = 0, arenaFrames = 0;   // frames the scratch arrays / the pyramid and blur arenas hold
    int rot = 0;                     // slot of the next sub-chunk
    hipEvent_t userReady = nullptr;  // rumi_orb_wait_event: the sub-chunks of the next resident-queue call start behind it
    bool lastResident = false;       // the previous batched call ran in the resident-queue arrangement
    hipStream_t part0Stream = nullptr, part0Side = nullptr;
    hipEvent_t evPart0Join = nullptr, evSide0Fork = nullptr, evSide0Join = nullptr;
    hipEvent_t evPartFork = nullptr, evPartJoin[kMaxParts - 1] = {nullptr}, evSideFork[kMaxParts - 1] = {nullptr}, evSideJoin[kMaxParts - 1] = {nullptr};
    hipEvent_t evFork = nullptr, evJoin = nullptr, evB0 = nullptr, evB1 = nullptr;
};

static int set_geometry(RumiOrb *h, int w, int hgt) {
    if (h->gw == w && h->gh == hgt) return RUMI_OK;
    std::vector<LevelGeom> g;
    long long arena; int cells, cand, cellCand;
    if (!make_geometry(h->tab, w, hgt, g, &arena, &cells, &cand, &cellCand)) {
        g_lastError = "image too small for the FAST cell grid at some pyramid level, or cell larger than the LDS tile";
        return RUMI_E_INVALID;
    }
    if (arena > h->capArena || cells > h->capCells || cand > h->capCand || cellCand > h->capCellCand) {
        g_lastError = "image larger than the handle's max_width x max_height arenas";
        return RUMI_E_CAPACITY;
    }
    DevParams &P = h->hP;
    std::memset(&P, 0, sizeof P);
    P.nlevels = h->tab.nlevels; P.totalCells = cells; P.maxCellCand = h->capCellCand; P.totalCand = h->capCand;
    P.iniTh = std::min(std::max(h->cfg.ini_th_fast, 0), 255);   // cv::FAST clamps its threshold
    P.minTh = std::min(std::max(h->cfg.min_th_fast, 0), 255);
    P.arenaStride = h->capArena;
    for (int i = 0; i < 16; i++) P.umax[i] = h->tab.umax[i];
    std::vector<int16_t> coef;
    std::vector<RowTap> rowTab;
    for (int l = 0; l < P.nlevels; l++) {
        DevLevel &D = P.lv[l];
        const LevelGeom &G = g[l];
        D.w = G.w; D.h = G.h; D.pitch = G.pitch; D.off = G.off;
        D.nCols = G.nCols; D.nRows = G.nRows; D.wCell = G.wCell; D.hCell = G.hCell;
        D.cellBase = G.cellBase; D.nCells = G.nCells; D.maxBX = G.maxBX; D.maxBY = G.maxBY;
        D.nfeat = G.nfeat; D.scale = G.scale;
        D.patchSize = (float)(int)(kPatchSize * G.scale);
        D.candCap = std::min(G.candCap, 65535);
        D.coefX = D.coefXT = D.coefY = 0; D.xmax = D.xmaxFast = G.w; D.rowTab = 0;
        if (l > 0) {
            std::vector<int16_t> ofs, taps;
            int dmax;
            make_resize_axis(g[l - 1].w, G.w, true, ofs, taps, &dmax);
            D.coefX = (int)coef.size(); D.xmax = dmax;
            D.xmaxFast = dmax == G.w ? G.w + 3 : dmax;
            for (int k = 0; k < 4; k++) { ofs.push_back(ofs[G.w - 1]); taps.push_back(taps[2 * G.w - 2]); taps.push_back(taps[2 * G.w - 1]); }
            coef.insert(coef.end(), ofs.begin(), ofs.end());
            D.coefXT = (int)coef.size();
            coef.insert(coef.end(), taps.begin(), taps.end());
            make_resize_axis(g[l - 1].h, G.h, false, ofs, taps, &dmax);
            D.coefY = (int)coef.size();
            coef.insert(coef.end(), ofs.begin(), ofs.end());
            coef.insert(coef.end(), taps.begin(), taps.end());
            // per output row: the two clamped source rows (cv clips the ROW indices when it fetches them) and the taps << 16
            D.rowTab = (int)rowTab.size();
            const int sh = g[l - 1].h;
            for (int oy = 0; oy < G.h; oy++) {
                const int sy = ofs[oy];
                const int sy0 = sy >= 0 ? (sy < sh ? sy : sh - 1) : 0, sy1r = sy + 1, sy1 = sy1r >= 0 ? (sy1r < sh ? sy1r : sh - 1) : 0;
                rowTab.push_back(RowTap{sy0, sy1, (uint32_t)taps[oy * 2] << 16, (uint32_t)taps[oy * 2 + 1] << 16});
            }
        }
    }
    if ((int)coef.size() > h->capCoef) { g_lastError = "resize table capacity"; return RUMI_E_CAPACITY; }
    HIP_TRY(hipMemcpy(h->dP, &P, sizeof P, hipMemcpyHostToDevice));
    if ((int)rowTab.size() > h->capRowTab) { g_lastError = "resize row table capacity"; return RUMI_E_CAPACITY; }
    if (!coef.empty()) HIP_TRY(hipMemcpy(h->dCoef, coef.data(), coef.size() * sizeof(int16_t), hipMemcpyHostToDevice));
    if (!rowTab.empty()) HIP_TRY(hipMemcpy(h->dRowTab, rowTab.data(), rowTab.size() * sizeof(RowTap), hipMemcpyHostToDevice));
    // ---- regions of the one-launch pyramid (k_pyramid_tiles): an even partition of the TOP level into tiles of about kPyrTX x kPyrTY pixels (16 x 8: 14.3 us for one 640 x 480 frame; 32 x 16: 19.4, 16 x 16: 16.5, 8 x 8: 15.1); going down, a tile's region of
    // level l - 1 is the hull of what its region of level l reads (first tap column .. second tap column, first .. second source row) and of its
    // share of an even partition of level l - 1 (every pixel of every level belongs to some tile); x ranges are widened to multiples of 4 (the
    // kernels store dwords; the tables carry 4 padded columns).  Level 0's "region" is the window of the frame the tile reads.
    // Two tile sets: small tiles (16 x 8 of the top level) for calls of a few frames, where the dependent chain is what counts (216 workgroups for
    // one 640 x 480 frame); large tiles (RUMI_PYR_BATCH_TILE, default 48 x 16) for batches (opt-in, RUMI_PYRAMID_TILES=3: measured slower, see extract_async_impl), where the chip is full and what counts is the recomputed
    // border (~1.2x the pixels instead of 1.8x) and the traffic: the frame is read once and every level written once, no level is read back.
    auto build_tiles = [&](int kPyrTX, int kPyrTY, int set) -> int {
        h->nPyrTiles[set] = 0; h->pyrBuf[set] = 0;
        if (P.nlevels < 2) return RUMI_OK;
        const int top = P.nlevels - 1;
        const int ntx = (P.lv[top].w + kPyrTX - 1) / kPyrTX, nty = (P.lv[top].h + kPyrTY - 1) / kPyrTY;
        std::vector<PyrTile> tiles((size_t)ntx * nty);
        int bufMax = 0, tabMax = 0, dimMax = 0, winRows = 0, winCols = 0;
        auto up4 = [](int x) { return (x + 3) & ~3; };
        for (int ty = 0; ty < nty; ty++)
            for (int tx = 0; tx < ntx; tx++) {
                PyrTile &T = tiles[(size_t)ty * ntx + tx];
                std::memset(&T, 0, sizeof T);
                int x0 = 0, x1 = 0, y0 = 0, y1 = 0, tab = 0;
                for (int l = top; l >= 0; l--) {
                    const DevLevel &D = P.lv[l];
                    // own share of level l
                    int ox0 = (int)((long long)D.w * tx / ntx), ox1 = (int)((long long)D.w * (tx + 1) / ntx);
                    int oy0 = (int)((long long)D.h * ty / nty), oy1 = (int)((long long)D.h * (ty + 1) / nty);
                    if (l < top) {
                        // what level l + 1's region [x0, x1) x [y0, y1) reads of level l
                        const DevLevel &U = P.lv[l + 1];
                        const int16_t *xofs = coef.data() + U.coefX;
                        const int nx0 = xofs[x0], nx1 = std::min(D.w, (int)xofs[x1 - 1] + 2);
                        const int ny0 = rowTab[(size_t)U.rowTab + y0].r0, ny1 = rowTab[(size_t)U.rowTab + y1 - 1].r1 + 1;
                        if (l == 0) { ox0 = nx0; ox1 = nx1; oy0 = ny0; oy1 = ny1; }          // level 0 is only read
                        else { ox0 = std::min(ox0, nx0); ox1 = std::max(ox1, nx1); oy0 = std::min(oy0, ny0); oy1 = std::max(oy1, ny1); }
                    }
                    if (l > 0) { x0 = ox0 & ~3; x1 = up4(ox1); } else { x0 = ox0 & ~3; x1 = ox1; }    // (level 0: dword loads from an aligned column)
                    y0 = oy0; y1 = oy1;
                    T.x0[l] = (int16_t)x0; T.x1[l] = (int16_t)x1; T.y0[l] = (int16_t)y0; T.y1[l] = (int16_t)y1;
                    bufMax = std::max(bufMax, up4(x1 - x0) * (y1 - y0));
                    if (l > 0) tab += (x1 - x0) + (y1 - y0);            // the tile's slices of the column and row tables (8 bytes an entry)
                    if (l > 0) dimMax = std::max(dimMax, std::max(x1 - x0, y1 - y0));
                    if (l == 0) { winRows = std::max(winRows, y1 - y0); winCols = std::max(winCols, x1 - x0); }
                }
                tabMax = std::max(tabMax, tab);
            }
        bufMax = (bufMax + 15) & ~15;
        if (2 * bufMax + 8 * tabMax <= 60 * 1024 && tiles.size() <= 4096 && P.nlevels <= 8 && dimMax <= 256 && winRows <= 80 && winCols <= 256) {   // (the kernel's fixed shapes: orb_kernels.hip)
            if (h->dPyrTiles[set]) { (void)hipFree(h->dPyrTiles[set]); h->dPyrTiles[set] = nullptr; }
            HIP_TRY(hipMalloc((void **)&h->dPyrTiles[set], tiles.size() * sizeof(PyrTile)));
            HIP_TRY(hipMemcpy(h->dPyrTiles[set], tiles.data(), tiles.size() * sizeof(PyrTile), hipMemcpyHostToDevice));
            h->nPyrTiles[set] = (int)tiles.size(); h->pyrBuf[set] = bufMax; h->pyrTab[set] = tabMax;
        }
        return RUMI_OK;
    };
    {
        int rcT = build_tiles(16, 8, 0);
        if (rcT != RUMI_OK) return rcT;
        int btx = 48, bty = 16;
        if (const char *e = std::getenv("RUMI_PYR_BATCH_TILE")) { if (std::sscanf(e, "%dx%d", &btx, &bty) != 2 || btx < 4 || bty < 2) { btx = 48; bty = 16; } }
        if ((rcT = build_tiles(btx, bty, 1)) != RUMI_OK) return rcT;
    }
    h->octLds = octree_lds_for(P);
    if (h->octLds > 160 * 1024) { g_lastError = "nfeatures too large for the LDS-resident quadtree node pool"; return RUMI_E_INVALID; }
    h->gw = w; h->gh = hgt;
    return RUMI_OK;
}

extern "C" int rumi_orb_tables(const RumiOrbConfig *cfg, float *scale, float *inv_scale, float *sigma2,
                               float *inv_sigma2, int32_t *features_per_level, int32_t *umax16) {
    if (!cfg || cfg->nlevels < 1 || cfg->nlevels > kMaxLevels) return RUMI_E_INVALID;
    OrbTables t = make_tables(cfg->nfeatures, cfg->scale_factor, cfg->nlevels);
    for (int i = 0; i < t.nlevels; i++) {
        if (scale) scale[i] = t.scale[i];
        if (inv_scale) inv_scale[i] = t.invScale[i];
        if (sigma2) sigma2[i] = t.sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = t.invSigma2[i];
        if (features_per_level) features_per_level[i] = t.featuresPerLevel[i];
    }
    if (umax16) for (int i = 0; i < 16; i++) umax16[i] = t.umax[i];
    return RUMI_OK;
}

extern "C" void rumi_orb_destroy(RumiOrb *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->pending) (void)hipStreamSynchronize(h->pendingStream);
    void *dev[] = {h->dP, h->dCoef, h->dRowTab, h->dPyrTiles[0], h->dPyrTiles[1], h->dIn, h->dL0, h->dPyr, h->dBlur, h->dCellBuf, h->dCellCnt, h->dCand, h->dLevelStart,
                   h->dSelPacked, h->dSelMeta, h->dSelCount, h->dKp, h->dDesc, h->dCounts,
                   h->dOwner, h->dSelLevel, h->dSelLevelCnt, h->dErr};
    for (void *p : dev) if (p) (void)hipFree(p);
    void *pin[] = {h->hErr};
    for (void *p : pin) if (p) (void)hipHostFree(p);
    for (auto &e : h->ev) if (e) (void)hipEventDestroy(e);
    if (h->evFork) (void)hipEventDestroy(h->evFork);
    if (h->evJoin) (void)hipEventDestroy(h->evJoin);
    if (h->evB0) (void)hipEventDestroy(h->evB0);
    if (h->evB1) (void)hipEventDestroy(h->evB1);
    if (h->sideStream) (void)hipStreamDestroy(h->sideStream);
    for (auto &ps : h->partStream) if (ps) (void)hipStreamDestroy(ps);
    for (auto &ps : h->partSide) if (ps) (void)hipStreamDestroy(ps);
    for (auto &e : h->evPartJoin) if (e) (void)hipEventDestroy(e);
    for (auto &e : h->evSideFork) if (e) (void)hipEventDestroy(e);
    for (auto &e : h->evSideJoin) if (e) (void)hipEventDestroy(e);
    if (h->evPartFork) (void)hipEventDestroy(h->evPartFork);
    if (h->part0Stream) (void)hipStreamDestroy(h->part0Stream);
    if (h->part0Side) (void)hipStreamDestroy(h->part0Side);
    for (hipEvent_t e : {h->evPart0Join, h->evSide0Fork, h->evSide0Join}) if (e) (void)hipEventDestroy(e);
    if (h->dHostIn) (void)hipFree(h->dHostIn);
    for (auto &p : h->hFeed) if (p) (void)hipHostFree(p);
    for (auto &e : h->evFeed) if (e) (void)hipEventDestroy(e);
    if (h->copyStream) (void)hipStreamDestroy(h->copyStream);
    if (h->copyStream2) (void)hipStreamDestroy(h->copyStream2);
    if (h->hIn) (void)hipHostFree(h->hIn);
    if (h->hOut1) (void)hipHostFree(h->hOut1);
    if (h->dOut1) (void)hipFree(h->dOut1);
    delete h;
}

// The per-frame device arrays: candidate / quadtree scratch for `scratch` frames, pyramid and blurred levels for `arena` frames.  Called by
// rumi_orb_create and again by rumi_orb_set_resident_queue when the slots of the resident queue need more than the handle was created with.
static int alloc_frame_arenas(RumiOrb *h, size_t scratch, size_t arena) {
    void *old[] = {h->dPyr, h->dBlur, h->dCellBuf, h->dCellCnt, h->dCand, h->dLevelStart, h->dSelPacked, h->dSelMeta, h->dSelCount, h->dOwner, h->dSelLevel, h->dSelLevelCnt};
    for (void *p : old) if (p) (void)hipFree(p);
    h->dPyr = h->dBlur = nullptr; h->dCellBuf = nullptr; h->dCellCnt = nullptr; h->dCand = nullptr; h->dLevelStart = nullptr; h->dSelPacked = h->dSelMeta = nullptr;
    h->dSelCount = nullptr; h->dOwner = nullptr; h->dSelLevel = nullptr; h->dSelLevelCnt = nullptr;
    const size_t C = scratch;
    int rc;
    h->scratchFrames = 0; h->arenaFrames = 0;          // until everything below has succeeded the handle has NO arenas (extract refuses to run: RUMI_E_CAPACITY)
#define TRY_A(x) if ((rc = (x)) != RUMI_OK) return rc;
    TRY_A(dev_alloc(&h->dPyr, (size_t)h->capArena * arena));
    TRY_A(dev_alloc(&h->dBlur, (size_t)h->capArena * arena));
    TRY_A(dev_alloc(&h->dCellBuf, C * h->capCells * h->capCellCand));
    TRY_A(dev_alloc(&h->dCellCnt, C * h->capCells));
    TRY_A(dev_alloc(&h->dCand, C * h->capCand));
    TRY_A(dev_alloc(&h->dLevelStart, C * (kMaxLevels + 1)));
    TRY_A(dev_alloc(&h->dSelPacked, C * h->capSel));
    TRY_A(dev_alloc(&h->dSelMeta, C * h->capSel));
    TRY_A(dev_alloc(&h->dSelCount, C));
    TRY_A(dev_alloc(&h->dOwner, C * h->capCand));
    TRY_A(dev_alloc(&h->dSelLevel, C * h->cfg.nlevels * h->selLevelCap));
    TRY_A(dev_alloc(&h->dSelLevelCnt, C * h->cfg.nlevels));
#undef TRY_A
    h->scratchFrames = (int)scratch; h->arenaFrames = (int)arena;
    return RUMI_OK;
}

extern "C" int rumi_orb_create(const RumiOrbConfig *cfg, RumiOrb **out) {
    if (!out) return RUMI_E_INVALID;
    *out = nullptr;
    if (!cfg || cfg->nlevels < 1 || cfg->nlevels > kMaxLevels || cfg->nfeatures < 1 || cfg->max_batch < 1 ||
        cfg->max_width < 1 || cfg->max_height < 1 || !(cfg->scale_factor > 1.0f) || cfg->blur_variant < 0 || cfg->blur_variant > 1) {
        g_lastError = "invalid RumiOrbConfig";
        return RUMI_E_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        g_lastError = "no HIP device visible: librumi_hip has no CPU fallback";
        return RUMI_E_NO_DEVICE;
    }
    RumiOrb *h = new RumiOrb();
    h->cfg = *cfg;
    if (cfg->device >= 0) { h->device = cfg->device; }
    else if (hipGetDevice(&h->device) != hipSuccess) h->device = 0;
    if (hipSetDevice(h->device) != hipSuccess) { delete h; g_lastError = "hipSetDevice failed"; return RUMI_E_NO_DEVICE; }
    h->tab = make_tables(cfg->nfeatures, cfg->scale_factor, cfg->nlevels);
    h->hostThreads = cfg->host_threads > 0 ? cfg->host_threads
                                           : (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    std::vector<LevelGeom> g;
    int cells, cand, cellCand;
    if (!make_geometry(h->tab, cfg->max_width, cfg->max_height, g, &h->capArena, &cells, &cand, &cellCand)) {
        delete h;
        g_lastError = "max_width x max_height too small for the FAST cell grid at some level";
        return RUMI_E_INVALID;
    }
    int coefN = 0;
    for (int l = 1; l < cfg->nlevels; l++) coefN += 3 * (g[l].w + g[l].h);
    h->capCells = cells + cells / 8 + 16;            // slack: smaller images can have slightly different grids
    h->capCellCand = cellCand + cellCand / 2 + 16;
    int candSum = 0;
    for (int l = 0; l < cfg->nlevels; l++) candSum += std::min(g[l].candCap, 65535);
    h->capCand = candSum + 64;
    h->capCoef = coefN + 64 * cfg->nlevels;
    h->capRowTab = 0;
    for (int l = 1; l < cfg->nlevels; l++) h->capRowTab += g[l].h + 8;
    h->capSel = cfg->nfeatures + 4 * cfg->nlevels + 64;   // the quadtree may return a few more than N per level
    int maxN = 0;
    for (int l = 0; l < cfg->nlevels; l++) maxN = std::max(maxN, h->tab.featuresPerLevel[l]);
    h->selLevelCap = maxN + 4 * 16 + 8;
    // scratch arenas: frames of one pass, rounded up to a multiple of 12 so that 2, 3 or 4 equal slots hold ceil(frames / parts) each
    const size_t B = (size_t)cfg->max_batch, C = (size_t)scratch_frames(cfg->max_batch);
    int rc = RUMI_OK;
#define TRY_ALLOC(x) if ((rc = (x)) != RUMI_OK) { rumi_orb_destroy(h); return rc; }
    TRY_ALLOC(dev_alloc(&h->dP, 1));
    TRY_ALLOC(dev_alloc(&h->dCoef, (size_t)h->capCoef));
    TRY_ALLOC(dev_alloc(&h->dRowTab, (size_t)std::max(h->capRowTab, 1)));
    TRY_ALLOC(dev_alloc(&h->dIn, (size_t)((cfg->max_width + 3) & ~3) * cfg->max_height));
    TRY_ALLOC(alloc_frame_arenas(h, C, B));
    TRY_ALLOC(dev_alloc(&h->dKp, (size_t)h->capSel));
    TRY_ALLOC(dev_alloc(&h->dDesc, (size_t)h->capSel * 32));
    TRY_ALLOC(dev_alloc(&h->dCounts, 2));
    TRY_ALLOC(dev_alloc(&h->dOut1, (size_t)16 + (size_t)h->capSel * 60));
    if (hipHostMalloc((void **)&h->hIn, (size_t)((cfg->max_width + 3) & ~3) * cfg->max_height, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&h->hOut1, (size_t)16 + (size_t)h->capSel * 60, hipHostMallocDefault) != hipSuccess) {
        rumi_orb_destroy(h); g_lastError = "pinned staging"; return RUMI_E_NO_DEVICE;
    }
    TRY_ALLOC(dev_alloc(&h->dErr, 1));
    TRY_ALLOC(pin_alloc(&h->hErr, 1));
    if (hipHostGetDevicePointer((void **)&h->dhOut1, h->hOut1, 0) != hipSuccess || hipHostGetDevicePointer((void **)&h->dhErr, h->hErr, 0) != hipSuccess) {
        (void)hipGetLastError();
        h->dhOut1 = nullptr; h->dhErr = nullptr;            // (no device view of the pinned blocks: the copies stay)
    }
    if (hipHostGetDevicePointer((void **)&h->dhIn, h->hIn, 0) != hipSuccess) { (void)hipGetLastError(); h->dhIn = nullptr; }
#undef TRY_ALLOC
    for (auto &e : h->ev)
        if (hipEventCreate(&e) != hipSuccess) { rumi_orb_destroy(h); g_lastError = "hipEventCreate"; return RUMI_E_NO_DEVICE; }
    for (int i = 0; i < RumiOrb::kMaxParts - 1; ++i)
        if (hipStreamCreateWithFlags(&h->partStream[i], hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&h->partSide[i], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&h->evSideFork[i], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&h->evSideJoin[i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->evPartJoin[i], hipEventDisableTiming) != hipSuccess) { rumi_orb_destroy(h); g_lastError = "part stream"; return RUMI_E_NO_DEVICE; }
    if (hipEventCreateWithFlags(&h->evPartFork, hipEventDisableTiming) != hipSuccess) { rumi_orb_destroy(h); g_lastError = "part event"; return RUMI_E_NO_DEVICE; }
    if (hipStreamCreateWithFlags(&h->part0Stream, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&h->part0Side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->evPart0Join, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&h->evSide0Fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->evSide0Join, hipEventDisableTiming) != hipSuccess) { rumi_orb_destroy(h); g_lastError = "part stream 0"; return RUMI_E_NO_DEVICE; }
    if (hipStreamCreateWithFlags(&h->sideStream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->evFork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->evJoin, hipEventDisableTiming) != hipSuccess || hipEventCreate(&h->evB0) != hipSuccess ||
        hipEventCreate(&h->evB1) != hipSuccess) { rumi_orb_destroy(h); g_lastError = "side stream"; return RUMI_E_NO_DEVICE; }
    *out = h;
    return RUMI_OK;
}

// Frames per sub-chunk of the resident queue.  256 since the end of round 3 (64 before): with the latency-bound kernels of a sub-chunk's chain
// shortened (compaction, quadtree) larger launches win at every call size (1024-frame steps +3 %, 128-frame calls +5 %); RUMI_RESIDENT_SUB overrides.
static int resident_sub_frames() {
    static const int v = std::getenv("RUMI_RESIDENT_SUB") ? std::max(1, std::atoi(std::getenv("RUMI_RESIDENT_SUB"))) : 256;
    return v;
}
extern "C" int rumi_orb_set_profiling(RumiOrb *h, int32_t on) {
    if (!h) return RUMI_E_INVALID;
    h->profiling = on != 0;
    return RUMI_OK;
}
extern "C" int rumi_orb_set_resident_queue(RumiOrb *h, int32_t on) {
    if (!h) return RUMI_E_INVALID;
    if (h->pending) { const int rc = rumi_orb_sync(h); if (rc != RUMI_OK) return rc; }
    if (on) {
        // `on` slots (1: the default of four) of up to 256 frames each: their scratch ranges and their ranges of the pyramid / blur arenas
        const int slots = on == 1 ? 4 : std::min(std::max(on, 2), (int)RumiOrb::kMaxParts);
        h->residentSlots = slots;
        const int need = (slots * std::min(resident_sub_frames(), h->cfg.max_batch) + 23) / 24 * 24;
        if (h->scratchFrames < need || h->arenaFrames < need) {
            HIP_TRY(hipSetDevice(h->device));
            HIP_TRY(hipDeviceSynchronize());
            const size_t oldS = (size_t)h->scratchFrames, oldA = (size_t)h->arenaFrames;
            int rc = alloc_frame_arenas(h, std::max(oldS, (size_t)need), std::max(oldA, (size_t)need));
            if (rc != RUMI_OK) {                             // back to the sizes the handle had (those fitted before); if even that fails the handle refuses to extract
                h->residentQueue = false;
                (void)alloc_frame_arenas(h, oldS, oldA);
                g_lastError = "resident queue: device arenas";
                return rc;
            }
            h->lastFrames = 0; h->tapValid = false;
        }
    }
    h->residentQueue = on != 0;
    return RUMI_OK;
}
extern "C" int rumi_orb_wait_event(RumiOrb *h, void *hip_event) {
    if (!h) return RUMI_E_INVALID;
    h->userReady = (hipEvent_t)hip_event;
    return RUMI_OK;
}
extern "C" int rumi_orb_stage_ms(RumiOrb *h, float ms[8]) {
    if (!h || !ms) return RUMI_E_INVALID;
    for (int i = 0; i < 8; i++) ms[i] = h->stageMs[i];
    return RUMI_OK;
}

extern "C" int rumi_orb_sync(RumiOrb *h) {
    if (!h) return RUMI_E_INVALID;
    if (!h->pending) return RUMI_OK;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->pendingStream));
    h->pending = false;
    const int err = *h->hErr;
    if (err & 16) { g_lastError = "FAST candidate capacity exceeded (more than 65535 in one level)"; return RUMI_E_CAPACITY; }
    if (err & 1) { g_lastError = "aspect ratio gives 0 or more than 16 quadtree roots"; return RUMI_E_INVALID; }
    if (err & 2) { g_lastError = "quadtree node pool exhausted"; return RUMI_E_INVALID; }
    if (err & (4 | 8)) { g_lastError = "more key-points than the handle's selection capacity"; return RUMI_E_CAPACITY; }
    return RUMI_OK;
}

extern "C" int rumi_orb_extract_batch_device(RumiOrb *h, const void *d_imgs, int32_t nframes, int32_t w, int32_t hgt,
                                             int32_t stride, int64_t frame_stride, int32_t lap0, int32_t lap1,
                                             void *d_kp, void *d_desc, void *d_counts, int32_t cap, void *hip_stream) {
    const int rc = rumi_orb_extract_batch_device_async(h, d_imgs, nframes, w, hgt, stride, frame_stride, lap0, lap1, d_kp, d_desc, d_counts, cap, hip_stream);
    if (rc != RUMI_OK) { if (h && h->pending) (void)rumi_orb_sync(h); return rc; }
    return rumi_orb_sync(h);
}

// Output addressing of one call: frame f's key-points start at kp + f * kpStride (bytes), likewise descriptors and the {n, monoIndex} pair.
// The three-array form has strides cap * 28 / cap * 32 / 8; the record form (one all-gather payload) has the record size for all three.
struct OutLayout { void *kp; long long kpStride; void *desc; long long descStride; void *counts; long long countsStride; };

static int extract_async_impl(RumiOrb *h, const void *d_imgs, int32_t nframes, int32_t w, int32_t hgt, int32_t stride, int64_t frame_stride,
                              int32_t lap0, int32_t lap1, const OutLayout &out, int32_t cap, void *hip_stream) {
    void *d_kp = out.kp, *d_desc = out.desc, *d_counts = out.counts;
    if (!h || !d_imgs || !d_kp || !d_desc || !d_counts || nframes < 1 || cap < 1 || stride < w) {
        g_lastError = "rumi_orb_extract_batch_device: bad argument";
        return RUMI_E_INVALID;
    }
    if (w <= 0 || hgt <= 0) return RUMI_E_EMPTY;
    if (nframes > h->cfg.max_batch) { g_lastError = "nframes > max_batch"; return RUMI_E_CAPACITY; }
    if (h->scratchFrames <= 0 || h->arenaFrames <= 0) { g_lastError = "the handle has no device arenas (an earlier rumi_orb_set_resident_queue failed to allocate them)"; return RUMI_E_CAPACITY; }
    HIP_TRY(hipSetDevice(h->device));
    int rc;
    if (h->pending && (h->gw != w || h->gh != hgt) && (rc = rumi_orb_sync(h)) != RUMI_OK) return rc;   // new tables must not overtake running kernels
    rc = set_geometry(h, w, hgt);
    if (rc != RUMI_OK) return rc;
    hipStream_t st = (hipStream_t)hip_stream;
    // calls not yet waited for share the scratch arenas in stream order: another stream (or the profiled path) waits for them first
    if (h->pending && (st != h->pendingStream || h->profiling)) { rc = rumi_orb_sync(h); if (rc != RUMI_OK) return rc; }
    const DevParams &P = h->hP;
    // Level 0 is read where the caller has it, as aligned dwords.  Frames whose base, pitch or frame stride is not a multiple of 4 are
    // first copied into an aligned staging arena (the only case that costs a copy).
    bool stagedL0 = false;       // the frames were copied on `st`: the slot streams of a resident call must then wait for `st` (below)
    if ((reinterpret_cast<uintptr_t>(d_imgs) & 3) || (stride & 3) || (frame_stride & 3)) {
        stagedL0 = true;
        const int wp = (w + 3) & ~3;
        if (!h->dL0) HIP_TRY(hipMalloc((void **)&h->dL0, (size_t)((h->cfg.max_width + 3) & ~3) * h->cfg.max_height * h->cfg.max_batch));
        for (int f = 0; f < nframes; f++)
            HIP_TRY(hipMemcpy2DAsync(h->dL0 + (size_t)f * wp * hgt, wp, (const uint8_t *)d_imgs + (long long)f * frame_stride, stride, w, hgt,
                                     hipMemcpyDeviceToDevice, st));
        d_imgs = h->dL0; stride = wp; frame_stride = (int64_t)wp * hgt;
    }
    ImgSrc src{(const uint8_t *)d_imgs, frame_stride, stride, h->dPyr, h->dBlur};
    const bool prof = h->profiling;
    float acc[8] = {0};
    h->tapValid = false;

    // Stage A: pyramid + blur.  Levels depend on each other, frames do not.  The blur runs on a side stream next to FAST / quadtree and
    // joins before rBRIEF (stage times are taken with the same overlap the timed path has).
    // RUMI_SERIAL=1 (profiling aid): everything on the caller's stream, so that every kernel's duration is its stand-alone duration.
    static const bool serial = std::getenv("RUMI_SERIAL") != nullptr;
    // Batches of 64 frames and more are pipelined: sub-chunks of frames run pyramid -> FAST -> ... -> rBRIEF on up to 4 streams (sub-chunk j
    // on stream j % parts, in scratch slot j % parts), so the narrow launches of one sub-chunk (upper pyramid levels, compaction, quadtree:
    // latency-bound, few waves) sit beside the wide VALU-bound ones of the others.  Profiling and RUMI_SERIAL keep one stream.
    static const int envParts = std::getenv("RUMI_PARTS") ? std::atoi(std::getenv("RUMI_PARTS")) : 4;
    const int parts = (!prof && !serial) ? std::min(std::min(std::max(envParts, 1), (int)RumiOrb::kMaxParts), std::max(nframes / 32, 1)) : 1;
    // the call's error word starts at zero: a memset on the caller's stream, or -- a call that runs as one part on that stream (a handful of
    // frames: every dispatch counts) -- a store by the first pyramid kernel, which nothing that writes the word precedes
    bool clearInKernel = !h->pending && parts == 1 && !(h->residentQueue && !prof && !serial && !h->feed) && P.nlevels > 1;
    if (!h->pending && !clearInKernel) HIP_TRY(hipMemsetAsync(h->dErr, 0, sizeof(int32_t), st));
    if (h->userReady && !(h->residentQueue && !prof && !serial && !h->feed)) { HIP_TRY(hipStreamWaitEvent(st, h->userReady, 0)); h->userReady = nullptr; }
    // the streams and events of one sub-chunk slot: main stream, blur stream, fork / join of the blur
    struct Lane { hipStream_t s, bs; hipEvent_t fork, join; };
    const bool resident = h->residentQueue && !prof && !serial && !h->feed;
    auto lane_of = [&](int slot) -> Lane {
        if (slot) return {h->partStream[slot - 1], serial ? h->partStream[slot - 1] : h->partSide[slot - 1], h->evSideFork[slot - 1], h->evSideJoin[slot - 1]};
        if (resident) return {h->part0Stream, h->part0Side, h->evSide0Fork, h->evSide0Join};      // (no slot runs on the caller's stream)
        return {st, serial || prof ? st : h->sideStream, h->evFork, h->evJoin};        // profiling: the blur on the call's stream too, so that every stage time is a stand-alone duration
    };
    // A few frames (the Tracking thread's call): FAST and the blur go out as ONE launch on the main stream (k_fast_blur).  As two launches the
    // blur runs on the side stream, and the event that forks it stalls the main queue for ~20 us on this runtime: more than the blur takes.
    static const int envFuse = std::getenv("RUMI_FUSE_FAST_BLUR") ? std::atoi(std::getenv("RUMI_FUSE_FAST_BLUR")) : -1;
    const bool fuseBlur = !prof && !serial && (envFuse >= 0 ? envFuse != 0 : nframes < 16) && fast_blur_fusable(P);
    // A few frames: the pyramid in ONE launch (k_pyramid_tiles) instead of a launch per level
    static const int envTiles = std::getenv("RUMI_PYRAMID_TILES") ? std::atoi(std::getenv("RUMI_PYRAMID_TILES")) : -1;
    // RUMI_PYRAMID_TILES: unset = the small tiles for calls of up to 4 frames, a launch per level beyond; 0 = a launch per level for every call;
    // 1 = the small tiles for every call; 3 = small tiles up to 4 frames and the LARGE ones beyond.  (3 was MEASURED for batches in round 4 and is
    // not the default: 1024-frame steps run at 208-210 k fps with 32 x 16 / 48 x 16 top-level tiles against 234 k with the per-level launches -- a
    // workgroup walks seven levels between barriers and its threads idle on the small ones, which costs more than the saved re-reads bring.)
    const int tileSet = nframes <= 4 || envTiles == 1 ? 0 : 1;
    const bool tilePyramid = !prof && h->nPyrTiles[tileSet] > 0 &&
                             (tileSet == 0 ? (!serial && (envTiles >= 0 ? envTiles != 0 : true)) : envTiles == 3);
    bool copyL0 = false;
    auto stage_a = [&](const ImgSrc &ps, int n, const Lane &L) -> int {
        hipStream_t s = L.s;
        if (prof) HIP_TRY(hipEventRecord(h->ev[0], s));
        if (tilePyramid) {
            launch_pyramid_tiles(h->dP, ps, h->dCoef, h->dRowTab, h->dPyrTiles[tileSet], h->nPyrTiles[tileSet], h->pyrBuf[tileSet], h->pyrTab[tileSet], n, s, clearInKernel ? h->dErr : nullptr, copyL0);
            clearInKernel = false;
        } else
        for (int l = 1; l < P.nlevels; l++) {
            launch_resize(h->dP, P, ps, h->dCoef, h->dRowTab, l, n, s, clearInKernel ? h->dErr : nullptr);
            clearInKernel = false;
        }
        if (prof) HIP_TRY(hipEventRecord(h->ev[1], s));
        if (fuseBlur) return RUMI_OK;
        HIP_TRY(hipEventRecord(L.fork, s));
        HIP_TRY(hipStreamWaitEvent(L.bs, L.fork, 0));
        if (prof) HIP_TRY(hipEventRecord(h->evB0, L.bs));
        launch_blur(h->dP, P, ps, n, h->cfg.blur_variant, L.bs);
        if (prof) HIP_TRY(hipEventRecord(h->evB1, L.bs));
        HIP_TRY(hipEventRecord(L.join, L.bs));
        HIP_TRY(hipGetLastError());
        return RUMI_OK;
    };
    // rumi_orb_extract's frame, still in pinned host memory: copied to dIn first.  (RUMI_ORB_ZERO_COPY_IN=1: read in place by the one-launch pyramid,
    // which keeps a copy as the arena's level 0 for everything after it -- one dependent transfer less, but MEASURED SLOWER: the kernel's 550 KB of
    // window reads over PCIe take 25 us more than the 13 us copy + queue latency they replace: 124.8 against 99.8 us per call.  Off by default.)
    static const bool zcIn = std::getenv("RUMI_ORB_ZERO_COPY_IN") && std::atoi(std::getenv("RUMI_ORB_ZERO_COPY_IN")) != 0;
    bool l0FromHost = false;
    if (h->hostImagePending) {
        h->hostImagePending = false;
        if (zcIn && h->dhIn && tilePyramid && parts == 1 && !resident && nframes == 1) { l0FromHost = true; src.l0 = h->dhIn; }
        else HIP_TRY(hipMemcpyAsync(h->dIn, h->hIn, (size_t)stride * hgt, hipMemcpyHostToDevice, st));
    }
    if (parts == 1 && !resident) {
        if (h->feed && (rc = h->feed(nframes, st)) != RUMI_OK) return rc;
        copyL0 = l0FromHost;
        rc = stage_a(src, nframes, lane_of(0));
        copyL0 = false;
        if (rc != RUMI_OK) return rc;
        if (l0FromHost) {                                    // from here on level 0 is the arena's copy
            src.l0 = h->dPyr + P.lv[0].off; src.l0FrameStride = P.arenaStride; src.l0Pitch = P.lv[0].pitch;
            frame_stride = P.arenaStride; stride = P.lv[0].pitch;
        }
    }

    // FAST -> compaction -> quadtree -> orientation + descriptors for the frames [frame0, frame0 + n) of the batch on stream s, in the scratch
    // arenas from frame slot scr0 on (every scratch array is indexed by frame slot, so disjoint slot ranges can run on different streams)
    // (arena0 >= 0: the sub-chunk's pyramid / blurred levels live at frame position arena0 of the arenas instead of at frame0)
    auto run_part = [&](int frame0, int n, int scr0, const Lane &L, bool timed, bool withStageA, int arena0) -> int {
        hipStream_t s = L.s;
        ImgSrc ps = src;
        ps.l0 = src.l0 + (long long)frame0 * frame_stride;
        ps.pyr = src.pyr + (long long)(arena0 >= 0 ? arena0 : frame0) * P.arenaStride;
        ps.blur = src.blur + (long long)(arena0 >= 0 ? arena0 : frame0) * P.arenaStride;
        if (withStageA) { const int ra = stage_a(ps, n, L); if (ra != RUMI_OK) return ra; }
        uint32_t *cellBuf = h->dCellBuf + (size_t)scr0 * P.totalCells * P.maxCellCand;
        int32_t *cellCnt = h->dCellCnt + (size_t)scr0 * P.totalCells;
        uint32_t *candp = h->dCand + (size_t)scr0 * P.totalCand;
        int32_t *lvStart = h->dLevelStart + (size_t)scr0 * (kMaxLevels + 1);
        uint32_t *selLevel = h->dSelLevel + (size_t)scr0 * P.nlevels * h->selLevelCap;
        int32_t *selLevelCnt = h->dSelLevelCnt + (size_t)scr0 * P.nlevels;
        uint32_t *selPacked = h->dSelPacked + (size_t)scr0 * h->capSel, *selMeta = h->dSelMeta + (size_t)scr0 * h->capSel;
        if (timed) HIP_TRY(hipEventRecord(h->ev[3], s));
        if (fuseBlur) (void)launch_fast_blur(h->dP, P, ps, cellBuf, cellCnt, n, h->cfg.blur_variant, s);
        else launch_fast(h->dP, P, ps, cellBuf, cellCnt, n, s);
        if (timed) HIP_TRY(hipEventRecord(h->ev[4], s));
        launch_compact(h->dP, P, cellBuf, cellCnt, candp, lvStart, h->dErr, n, s);
        if (timed) HIP_TRY(hipEventRecord(h->ev[5], s));
        launch_octree(h->dP, P, candp, lvStart, h->dOwner + (size_t)scr0 * P.totalCand, selLevel, selLevelCnt, h->selLevelCap, h->dErr, n, h->octLds, s);
        // a few frames: the slot assignment (k_assemble) inside the descriptor kernel's prologue, one launch less on the dependent chain
        static const int envFuseA = std::getenv("RUMI_FUSE_ASSEMBLE") ? std::atoi(std::getenv("RUMI_FUSE_ASSEMBLE")) : -1;
        const bool fuseAssemble = !timed && !serial && (envFuseA >= 0 ? envFuseA != 0 : n <= 4) && (long long)((h->capSel + 7) / 8) * n <= 2048;
        int32_t *countsOut = (int32_t *)((uint8_t *)d_counts + (size_t)frame0 * out.countsStride);
        if (!fuseAssemble)
            launch_assemble(h->dP, selLevel, selLevelCnt, h->selLevelCap, lap0, lap1, selPacked, selMeta, h->dSelCount + scr0, h->capSel,
                            countsOut, out.countsStride, h->dErr, n, s, h->zeroCopyOut ? h->dhErr : nullptr);
        if (timed) HIP_TRY(hipEventRecord(h->ev[6], s));
        if (!fuseBlur) HIP_TRY(hipStreamWaitEvent(s, L.join, 0));   // join: rBRIEF reads the blurred levels
        if (fuseAssemble)
            launch_assemble_orient_desc(h->dP, ps, selLevel, selLevelCnt, h->selLevelCap, lap0, lap1, countsOut, out.countsStride, h->dErr,
                                        h->zeroCopyOut ? h->dhErr : nullptr, selPacked, selMeta, h->dSelCount + scr0, h->capSel, h->capSel,
                                        (RumiKeyPoint *)((uint8_t *)d_kp + (size_t)frame0 * out.kpStride), out.kpStride,
                                        (uint8_t *)d_desc + (size_t)frame0 * out.descStride, out.descStride, cap, n, s);
        else
        launch_orient_desc(h->dP, ps, selPacked, selMeta, h->dSelCount + scr0, h->capSel, h->capSel,
                           (RumiKeyPoint *)((uint8_t *)d_kp + (size_t)frame0 * out.kpStride), out.kpStride,
                           (uint8_t *)d_desc + (size_t)frame0 * out.descStride, out.descStride, cap, n, s);
        if (timed) HIP_TRY(hipEventRecord(h->ev[7], s));
        for (const auto &mr : h->mirror)
            if (mr.host) HIP_TRY(hipMemcpyAsync(mr.host + (size_t)frame0 * mr.row, mr.dev + (size_t)frame0 * mr.row, (size_t)n * mr.row, hipMemcpyDeviceToHost, s));
        return RUMI_OK;
    };
    if (resident) {
        // Resident queue: FOUR fixed slots (stream, blur stream, scratch range, pyramid / blur arena range), sub-chunks of at most 256 frames
        // dealt to the slots round-robin ACROSS calls (a 64-frame call takes one slot, the next call the next one).  Everything a sub-chunk
        // touches on the device belongs to its slot, so stream order alone keeps consecutive users of a slot apart: no sub-chunk waits for
        // the caller's stream or for another slot -- except after a rumi_orb_sync, whose reset of the error word is queued on `st`.
        constexpr int kMaxSlots = RumiOrb::kMaxParts;
        const int kSlots = h->residentSlots;
        const int slotFrames = h->scratchFrames / kSlots;
        const int cap64 = std::min(resident_sub_frames(), slotFrames);
        const int nsub = (nframes + cap64 - 1) / cap64, sub = (nframes + nsub - 1) / nsub;
        // (unaligned frames were staged into dL0 by copies queued on `st`: every slot stream this call touches waits for them.  The previous
        // call's readers of dL0 are behind `st` already: the caller's stream waited for that call's results at its end.)
        const bool fork = !h->pending || !h->lastResident || stagedL0;
        if (fork) HIP_TRY(hipEventRecord(h->evPartFork, st));
        bool touched[kMaxSlots] = {false, false, false, false, false, false, false, false};
        for (int j = 0, base = 0; base < nframes; j++, base += sub) {
            const int n = std::min(sub, nframes - base), slot = (h->rot + j) % kSlots;
            const Lane L = lane_of(slot);
            if (fork && !touched[slot]) HIP_TRY(hipStreamWaitEvent(L.s, h->evPartFork, 0));
            if (h->userReady && !touched[slot]) HIP_TRY(hipStreamWaitEvent(L.s, h->userReady, 0));
            touched[slot] = true;
            rc = run_part(base, n, slot * slotFrames, L, false, true, slot * slotFrames);
            if (rc != RUMI_OK) return rc;
            h->lastChunkBase = base; h->lastChunkFrames = n; h->lastChunkSlot = slot * slotFrames;
        }
        h->rot = (h->rot + nsub) % kSlots;
        h->userReady = nullptr;
        // Join: the caller's stream waits for the results of every sub-chunk (an event per slot, taken after the slot's last sub-chunk)
        for (int p = 0; p < kSlots; p++)
            if (touched[p]) {
                hipEvent_t e = p ? h->evPartJoin[p - 1] : h->evPart0Join;
                HIP_TRY(hipEventRecord(e, lane_of(p).s));
                HIP_TRY(hipStreamWaitEvent(st, e, 0));
            }
        HIP_TRY(hipGetLastError());
    } else if (parts > 1) {
        // equal sub-chunks: rounds of `parts` sub-chunks, as few rounds as the slots allow, no short tail
        const int slotFrames = h->scratchFrames / parts;
        static const int envSub = std::getenv("RUMI_SUBMAX") ? std::atoi(std::getenv("RUMI_SUBMAX")) : 1 << 30;
        const int subMax = std::max(1, std::min(std::min(slotFrames, 64), envSub));      // (64: the host path's transfer groups; the arenas may hold more since the resident queue grew them)
        const int rounds = (nframes + parts * subMax - 1) / (parts * subMax), sub = (nframes + parts * rounds - 1) / (parts * rounds);
        HIP_TRY(hipEventRecord(h->evPartFork, st));
        int used = 0;
        for (int j = 0, base = 0; base < nframes; j++, base += sub) {
            const int n = std::min(sub, nframes - base), slot = j % parts;
            const Lane L = lane_of(slot);
            if (j < parts && slot) HIP_TRY(hipStreamWaitEvent(L.s, h->evPartFork, 0));
            if (h->feed && (rc = h->feed(base + n, L.s)) != RUMI_OK) return rc;
            rc = run_part(base, n, slot * slotFrames, L, false, true, -1);
            if (rc != RUMI_OK) return rc;
            used = std::max(used, slot + 1);
            h->lastChunkBase = base; h->lastChunkFrames = n; h->lastChunkSlot = slot * slotFrames;
        }
        for (int p = 1; p < used; p++) {
            HIP_TRY(hipEventRecord(h->evPartJoin[p - 1], h->partStream[p - 1]));
            HIP_TRY(hipStreamWaitEvent(st, h->evPartJoin[p - 1], 0));
        }
        HIP_TRY(hipGetLastError());
    } else {
        // one stream: chunks of kChunk frames reuse the scratch arenas in stream order
        for (int base = 0; base < nframes; base += kChunk) {
            const int nf = std::min(kChunk, nframes - base);
            rc = run_part(base, nf, 0, lane_of(0), prof, false, -1);
            if (rc != RUMI_OK) return rc;
            HIP_TRY(hipGetLastError());
            if (prof) {
                float ms;
                HIP_TRY(hipStreamSynchronize(st));
                HIP_TRY(hipEventElapsedTime(&ms, h->ev[3], h->ev[4])); acc[1] += ms;
                HIP_TRY(hipEventElapsedTime(&ms, h->ev[4], h->ev[5])); acc[2] += ms;
                HIP_TRY(hipEventElapsedTime(&ms, h->ev[5], h->ev[6])); acc[4] += ms;
                HIP_TRY(hipEventElapsedTime(&ms, h->ev[6], h->ev[7])); acc[5] += ms;
            }
            h->lastChunkBase = base; h->lastChunkFrames = nf; h->lastChunkSlot = 0;
        }
    }
    // The call's error word (and, for the single-frame host API, its result block) follow the kernels on the stream; rumi_orb_sync waits
    // for them.  Nothing here blocks, so a caller can queue the next batch while this one runs.
    if (h->out1Bytes) HIP_TRY(hipMemcpyAsync(h->hOut1, h->dOut1, h->out1Bytes, hipMemcpyDeviceToHost, st));
    if (!h->zeroCopyOut) HIP_TRY(hipMemcpyAsync(h->hErr, h->dErr, sizeof(int32_t), hipMemcpyDeviceToHost, st));       // (zero-copy: k_assemble has published it)
    h->pending = true; h->pendingStream = st;
    if (prof) {
        float ms;
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipEventElapsedTime(&ms, h->ev[0], h->ev[1])); acc[0] = ms;
        HIP_TRY(hipEventElapsedTime(&ms, h->evB0, h->evB1)); acc[3] = ms;
        HIP_TRY(hipEventElapsedTime(&ms, h->ev[0], h->ev[7])); acc[6] = ms;
        for (int i = 0; i < 8; i++) h->stageMs[i] = acc[i];
    }
    h->lastSrc = src; h->lastFrames = nframes;
    h->lastResident = resident;
    if (resident) {      // the arenas hold the pyramids of the last sub-chunk of each slot only; the taps serve the call's last sub-chunk
        h->lastSrc.pyr = src.pyr + ((long long)h->lastChunkSlot - h->lastChunkBase) * P.arenaStride;
        h->lastSrc.blur = src.blur + ((long long)h->lastChunkSlot - h->lastChunkBase) * P.arenaStride;
    }
    h->lastKp = (RumiKeyPoint *)d_kp; h->lastKpStride = out.kpStride; h->lastOutCap = cap;
    h->lastCounts = (int32_t *)d_counts;
    return RUMI_OK;
}

extern "C" int rumi_orb_extract_batch_device_async(RumiOrb *h, const void *d_imgs, int32_t nframes, int32_t w, int32_t hgt,
                                                   int32_t stride, int64_t frame_stride, int32_t lap0, int32_t lap1,
                                                   void *d_kp, void *d_desc, void *d_counts, int32_t cap, void *hip_stream) {
    const OutLayout out{d_kp, (long long)cap * (long long)sizeof(RumiKeyPoint), d_desc, (long long)cap * 32, d_counts, 8};
    return extract_async_impl(h, d_imgs, nframes, w, hgt, stride, frame_stride, lap0, lap1, out, cap, hip_stream);
}

// One fixed-capacity record per frame, {int32 n; int32 monoIndex; RumiKeyPoint kp[cap]; uint8 desc[cap][32]} = 8 + 60 cap bytes: the payload of the
// rumination queue's single all-gather.  record_bytes >= that size and a multiple of 4.
extern "C" int rumi_orb_extract_batch_records_async(RumiOrb *h, const void *d_imgs, int32_t nframes, int32_t w, int32_t hgt,
                                                    int32_t stride, int64_t frame_stride, int32_t lap0, int32_t lap1,
                                                    void *d_records, int64_t record_bytes, int32_t cap, void *hip_stream) {
    if (!d_records || cap < 1 || record_bytes < 8 + 60ll * cap || (record_bytes & 3)) { g_lastError = "rumi_orb_extract_batch_records: bad record size"; return RUMI_E_INVALID; }
    uint8_t *r = (uint8_t *)d_records;
    const OutLayout out{r + 8, record_bytes, r + 8 + (size_t)cap * sizeof(RumiKeyPoint), record_bytes, r, record_bytes};
    return extract_async_impl(h, d_imgs, nframes, w, hgt, stride, frame_stride, lap0, lap1, out, cap, hip_stream);
}

// Host-resident batch: the rumination queue holds its frames as host cv::Mats (CloudImageSampler.cc:148-170).  The frames travel to the device
// in groups of 64 on a copy stream of their own, each group's extraction waits only for its own group, so the transfers run under the kernels
// of the groups before it.  Pinned sources (hipHostMalloc / hipHostRegister) are copied from where they lie; pageable ones pass through four
// pinned staging slots filled by the handle's host threads.
static int extract_batch_host_impl(RumiOrb *h, const uint8_t *const *imgs, int32_t nframes, int32_t w, int32_t hgt, int32_t stride,
                                   int32_t lap0, int32_t lap1, const OutLayout &out, int32_t cap, void *hip_stream, const std::function<int(hipStream_t)> &tail) {
    if (!h || !imgs || !out.kp || !out.desc || !out.counts || nframes < 1 || cap < 1 || stride < w) {
        g_lastError = "rumi_orb_extract_batch_host: bad argument";
        return RUMI_E_INVALID;
    }
    if (w <= 0 || hgt <= 0) return RUMI_E_EMPTY;
    if (nframes > h->cfg.max_batch) { g_lastError = "nframes > max_batch"; return RUMI_E_CAPACITY; }
    for (int f = 0; f < nframes; f++) if (!imgs[f]) { g_lastError = "rumi_orb_extract_batch_host: null frame"; return RUMI_E_INVALID; }
    HIP_TRY(hipSetDevice(h->device));
    int rc;
    if (h->pending && (rc = rumi_orb_sync(h)) != RUMI_OK) return rc;
    const int wp = (w + 3) & ~3;
    const size_t frameBytes = (size_t)wp * hgt;
    if (h->dHostInBytes < frameBytes * nframes) {
        if (h->dHostIn) HIP_TRY(hipFree(h->dHostIn));
        h->dHostIn = nullptr; h->dHostInBytes = 0;
        HIP_TRY(hipMalloc((void **)&h->dHostIn, frameBytes * h->cfg.max_batch));
        h->dHostInBytes = frameBytes * h->cfg.max_batch;
    }
    if (!h->copyStream) {
        HIP_TRY(hipStreamCreateWithFlags(&h->copyStream, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&h->copyStream2, hipStreamNonBlocking));
        for (auto &e : h->evFeed) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    // is the source pinned?  (one answer for the whole queue: the frames of a queue come from one allocator)
    hipPointerAttribute_t attr{};
    const bool pinned = hipPointerGetAttributes(&attr, imgs[0]) == hipSuccess && attr.type == hipMemoryTypeHost;
    (void)hipGetLastError();
    constexpr int G = RumiOrb::kFeedFrames, S = RumiOrb::kFeedSlots;
    if (!pinned && h->hFeedBytes < frameBytes * G) {
        for (auto &p : h->hFeed) { if (p) HIP_TRY(hipHostFree(p)); p = nullptr; }
        for (auto &p : h->hFeed) HIP_TRY(hipHostMalloc((void **)&p, (size_t)((h->cfg.max_width + 3) & ~3) * h->cfg.max_height * G, hipHostMallocDefault));
        h->hFeedBytes = (size_t)((h->cfg.max_width + 3) & ~3) * h->cfg.max_height * G;
    }
    hipStream_t st = (hipStream_t)hip_stream;
    int fed = 0, group = 0;                                    // frames already on their way, groups enqueued
    static const bool twoCopyStreams = std::getenv("RUMI_ONE_COPY_STREAM") == nullptr;
    h->feed = [&](int upto, hipStream_t s) -> int {
        while (fed < upto) {
            const int n = std::min(G, nframes - fed), slot = group % S;
            hipStream_t cs = (twoCopyStreams && (group & 1)) ? h->copyStream2 : h->copyStream;
            if (group >= S) HIP_TRY(hipEventSynchronize(h->evFeed[slot]));       // the slot's previous group has left the pinned buffer / its event is free again
            bool dense = pinned && stride == wp;                               // one buffer, frames back to back: one transfer per group
            for (int f = 1; dense && f < n; f++) dense = imgs[fed + f] == imgs[fed] + (size_t)f * frameBytes;
            if (dense) {
                HIP_TRY(hipMemcpyAsync(h->dHostIn + (size_t)fed * frameBytes, imgs[fed], frameBytes * n, hipMemcpyHostToDevice, cs));
            } else if (pinned) {
                for (int f = 0; f < n; f++)
                    HIP_TRY(hipMemcpy2DAsync(h->dHostIn + (size_t)(fed + f) * frameBytes, wp, imgs[fed + f], stride, w, hgt, hipMemcpyHostToDevice, cs));
            } else {
                uint8_t *dst = h->hFeed[slot];
                const int nt = std::max(1, std::min(h->hostThreads, n));
                auto work = [&](int t) {
                    for (int f = t; f < n; f += nt)
                        for (int y = 0; y < hgt; y++) std::memcpy(dst + (size_t)f * frameBytes + (size_t)y * wp, imgs[fed + f] + (size_t)y * stride, (size_t)w);
                };
                std::vector<std::thread> th;
                for (int t = 1; t < nt; t++) th.emplace_back(work, t);
                work(0);
                for (auto &x : th) x.join();
                HIP_TRY(hipMemcpyAsync(h->dHostIn + (size_t)fed * frameBytes, dst, frameBytes * n, hipMemcpyHostToDevice, cs));
            }
            HIP_TRY(hipEventRecord(h->evFeed[slot], cs));
            fed += n; group++;
        }
        // copies complete in order on each copy stream: waiting for the newest group of each covers every frame below `upto`
        HIP_TRY(hipStreamWaitEvent(s, h->evFeed[(group - 1) % S], 0));
        if (twoCopyStreams && group >= 2) HIP_TRY(hipStreamWaitEvent(s, h->evFeed[(group - 2) % S], 0));
        return RUMI_OK;
    };
    rc = extract_async_impl(h, h->dHostIn, nframes, w, hgt, wp, (int64_t)frameBytes, lap0, lap1, out, cap, hip_stream);
    h->feed = nullptr;
    if (rc != RUMI_OK) { if (h->pending) (void)rumi_orb_sync(h); return rc; }
    if (tail && (rc = tail(st)) != RUMI_OK) { (void)rumi_orb_sync(h); return rc; }
    return rumi_orb_sync(h);
}

extern "C" int rumi_orb_extract_batch_host(RumiOrb *h, const uint8_t *const *imgs, int32_t nframes, int32_t w, int32_t hgt, int32_t stride,
                                           int32_t lap0, int32_t lap1, void *d_kp, void *d_desc, void *d_counts, int32_t cap,
                                           RumiKeyPoint *h_kp, uint8_t *h_desc, int32_t *h_counts, void *hip_stream) {
    const OutLayout out{d_kp, (long long)cap * (long long)sizeof(RumiKeyPoint), d_desc, (long long)cap * 32, d_counts, 8};
    // host arrays: one copy each at the end; optionally (pinned arrays) every sub-chunk's rows behind that sub-chunk's kernels (run_part)
    // (MEASURED SLOWER for these three arrays, 4.4 MB per 64-frame sub-chunk: host frames in, everything out, 1024 frames a step: 101 k fps against 105-122 k
    // with one copy each at the end -- the copies hold the sub-chunk streams while the uploads are the bottleneck.  Only with RUMI_ORB_MIRROR=2.  The record
    // layout below gains 4 % from the same mechanism and has it by default.)
    static const bool mirrorOn = std::getenv("RUMI_ORB_MIRROR") && std::atoi(std::getenv("RUMI_ORB_MIRROR")) == 2;
    auto pinned = [](const void *p) {
        hipPointerAttribute_t attr{};
        const bool yes = mirrorOn && p && hipPointerGetAttributes(&attr, p) == hipSuccess && attr.type == hipMemoryTypeHost;
        (void)hipGetLastError();
        return yes;
    };
    const bool pk = h && d_kp && pinned(h_kp), pd = h && d_desc && pinned(h_desc), pc = h && d_counts && pinned(h_counts);
    if (pk) h->mirror[0] = {(uint8_t *)h_kp, (const uint8_t *)d_kp, out.kpStride};
    if (pd) h->mirror[1] = {h_desc, (const uint8_t *)d_desc, out.descStride};
    if (pc) h->mirror[2] = {(uint8_t *)h_counts, (const uint8_t *)d_counts, out.countsStride};
    const int rc = extract_batch_host_impl(h, imgs, nframes, w, hgt, stride, lap0, lap1, out, cap, hip_stream, [&](hipStream_t st) -> int {
        if (h_counts && !pc) HIP_TRY(hipMemcpyAsync(h_counts, d_counts, (size_t)nframes * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        if (h_kp && !pk) HIP_TRY(hipMemcpyAsync(h_kp, d_kp, (size_t)nframes * cap * sizeof(RumiKeyPoint), hipMemcpyDeviceToHost, st));
        if (h_desc && !pd) HIP_TRY(hipMemcpyAsync(h_desc, d_desc, (size_t)nframes * cap * 32, hipMemcpyDeviceToHost, st));
        return RUMI_OK;
    });
    if (h) for (auto &mr : h->mirror) mr = {nullptr, nullptr, 0};
    return rc;
}

// The host-resident queue with ONE record per frame as output (the all-gather payload, rumi_orb_extract_batch_records_async's layout); h_records:
// optional host copy of the nframes records.
extern "C" int rumi_orb_extract_batch_host_records(RumiOrb *h, const uint8_t *const *imgs, int32_t nframes, int32_t w, int32_t hgt, int32_t stride,
                                                   int32_t lap0, int32_t lap1, void *d_records, int64_t record_bytes, int32_t cap, uint8_t *h_records,
                                                   void *hip_stream) {
    if (!d_records || cap < 1 || record_bytes < 8 + 60ll * cap || (record_bytes & 3)) { g_lastError = "rumi_orb_extract_batch_host_records: bad record size"; return RUMI_E_INVALID; }
    uint8_t *r = (uint8_t *)d_records;
    const OutLayout out{r + 8, record_bytes, r + 8 + (size_t)cap * sizeof(RumiKeyPoint), record_bytes, r, record_bytes};
    // a pinned destination takes the records sub-chunk by sub-chunk behind the kernels (run_part); a pageable one (whose "asynchronous" copy would hold
    // the enqueuing thread) gets them in one copy at the end
    hipPointerAttribute_t attr{};
    static const bool mirrorOn = !(std::getenv("RUMI_ORB_MIRROR") && std::atoi(std::getenv("RUMI_ORB_MIRROR")) == 0);
    const bool pinnedOut = mirrorOn && h && h_records && hipPointerGetAttributes(&attr, h_records) == hipSuccess && attr.type == hipMemoryTypeHost;
    (void)hipGetLastError();
    if (pinnedOut) h->mirror[0] = {h_records, r, record_bytes};
    const int rc = extract_batch_host_impl(h, imgs, nframes, w, hgt, stride, lap0, lap1, out, cap, hip_stream, [&](hipStream_t st) -> int {
        if (h_records && !pinnedOut) HIP_TRY(hipMemcpyAsync(h_records, d_records, (size_t)nframes * record_bytes, hipMemcpyDeviceToHost, st));
        return RUMI_OK;
    });
    if (h) h->mirror[0] = {nullptr, nullptr, 0};
    return rc;
}

// The handle's pinned staging buffer for a w x hgt frame, for a caller that lets its camera driver / decoder write the frame there (a cv::Mat
// constructed on this memory): rumi_orb_extract called with this pointer and stride skips its staging copy.
extern "C" int rumi_orb_image_buffer(RumiOrb *h, int32_t w, int32_t hgt, uint8_t **buf, int32_t *stride) {
    if (!h || !buf || !stride) return RUMI_E_INVALID;
    if (w <= 0 || hgt <= 0 || w > h->cfg.max_width || hgt > h->cfg.max_height) { g_lastError = "rumi_orb_image_buffer: frame larger than the handle was created for"; return RUMI_E_CAPACITY; }
    *buf = h->hIn; *stride = (w + 3) & ~3;
    return RUMI_OK;
}

extern "C" int rumi_orb_extract(RumiOrb *h, const uint8_t *img, int32_t w, int32_t hgt, int32_t stride, int32_t lap0,
                                int32_t lap1, RumiKeyPoint *kp_out, uint8_t *desc_out, int32_t cap, int32_t *n_out,
                                int32_t *mono_out) {
    if (n_out) *n_out = 0;
    if (mono_out) *mono_out = -1;
    if (!h || !n_out || !mono_out) return RUMI_E_INVALID;
    if (!img || w <= 0 || hgt <= 0) return RUMI_E_EMPTY;            // operator() returns -1 on an empty image
    if (stride < w || w > h->cfg.max_width || hgt > h->cfg.max_height) { g_lastError = "image size"; return RUMI_E_INVALID; }
    HIP_TRY(hipSetDevice(h->device));
    // image -> pinned -> device (async), kernels, [counts | key-points | descriptors] -> pinned: one synchronisation in all
    const int wp = (w + 3) & ~3;                                     // rows padded so that level 0 can be read as aligned dwords
    if (!(img == h->hIn && stride == wp))                    // (a caller that captured straight into rumi_orb_image_buffer's memory has nothing to stage)
        for (int y = 0; y < hgt; y++) std::memcpy(h->hIn + (size_t)y * wp, img + (size_t)y * stride, (size_t)w);
    h->hostImagePending = true;                              // (extract_async_impl reads it in place or copies it: see there)
    // Results straight into pinned host memory: the kernels' output pointers are the device's view of hOut1 (k_assemble writes the counts and the
    // final error word, k_orient_desc key-points and descriptors), so the call ends with its last kernel -- no copy back, no second copy for the
    // error word (two dependent transfers of ~6 + 2 us with ~9 us of queue latency each).  RUMI_ORB_ZERO_COPY=0 keeps the copies (A/B measurements).
    static const bool zc = !(std::getenv("RUMI_ORB_ZERO_COPY") && std::atoi(std::getenv("RUMI_ORB_ZERO_COPY")) == 0);
    const bool zero = zc && h->dhOut1 && h->dhErr && !h->profiling;
    uint8_t *ob = zero ? h->dhOut1 : h->dOut1;
    int32_t *dC = reinterpret_cast<int32_t *>(ob);
    RumiKeyPoint *dK = reinterpret_cast<RumiKeyPoint *>(ob + 16);
    uint8_t *dD = ob + 16 + (size_t)h->capSel * sizeof(RumiKeyPoint);
    h->out1Bytes = zero ? 0 : (size_t)16 + (size_t)h->capSel * 60;
    h->zeroCopyOut = zero;
    if (zero) *h->hErr = 0;
    const int rc = rumi_orb_extract_batch_device(h, h->dIn, 1, w, hgt, wp, (int64_t)wp * hgt, lap0, lap1, dK, dD, dC, h->capSel, nullptr);
    h->out1Bytes = 0;
    h->zeroCopyOut = false;
    h->hostImagePending = false;
    if (rc != RUMI_OK) return rc;
    const int32_t *counts = reinterpret_cast<const int32_t *>(h->hOut1);
    *n_out = counts[0];
    *mono_out = counts[1];
    if (counts[0] > cap) { g_lastError = "kp_out/desc_out capacity"; return RUMI_E_CAPACITY; }
    if (counts[0] > 0) {
        if (!kp_out || !desc_out) return RUMI_E_INVALID;
        std::memcpy(kp_out, h->hOut1 + 16, (size_t)counts[0] * sizeof(RumiKeyPoint));
        std::memcpy(desc_out, h->hOut1 + 16 + (size_t)h->capSel * sizeof(RumiKeyPoint), (size_t)counts[0] * 32);
    }
    return RUMI_OK;
}

extern "C" int rumi_orb_pyramid_level(RumiOrb *h, int32_t frame, int32_t level, int32_t which, int32_t border,
                                      uint8_t *out, int32_t out_stride, int32_t *w_out, int32_t *h_out) {
    if (!h || h->lastFrames == 0 || frame < 0 || frame >= h->lastFrames || level < 0 || level >= h->hP.nlevels || border < 0)
        return RUMI_E_INVALID;
    if (h->lastResident && (frame < h->lastChunkBase || frame >= h->lastChunkBase + h->lastChunkFrames)) {
        g_lastError = "with a resident queue the arenas keep the pyramid of the call's last sub-chunk only";
        return RUMI_E_INVALID;
    }
    const DevLevel &L = h->hP.lv[level];
    if (w_out) *w_out = L.w;
    if (h_out) *h_out = L.h;
    if (!out) return RUMI_OK;
    if (out_stride < L.w + 2 * border) return RUMI_E_CAPACITY;
    HIP_TRY(hipSetDevice(h->device));
    if (h->pending) { const int rcs = rumi_orb_sync(h); if (rcs != RUMI_OK) return rcs; }
    // no border is stored; the 19-px border copyMakeBorder(..., BORDER_REFLECT_101) gives mvImagePyramid (ORBextractor.cc:1105-1108)
    // is synthesised here from the interior, which is the same pixels by definition
    if (border > (which ? 0 : kEdge)) { g_lastError = which ? "blurred levels carry no border" : "border larger than EDGE_THRESHOLD (19)"; return RUMI_E_INVALID; }
    // level 0 is the caller's frame itself (it must still be alive); the other levels and every blurred level come from the arenas
    const bool own = !which && level == 0;
    const uint8_t *srcp = own ? h->lastSrc.l0 + (long long)frame * h->lastSrc.l0FrameStride
                              : (which ? h->lastSrc.blur : h->lastSrc.pyr) + (long long)frame * h->hP.arenaStride + L.off;
    uint8_t *inner = out + (size_t)border * out_stride + border;
    HIP_TRY(hipMemcpy2D(inner, out_stride, srcp, own ? h->lastSrc.l0Pitch : L.pitch, L.w, L.h, hipMemcpyDeviceToHost));
    auto refl = [](int p, int n) { if (n == 1) return 0; while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p; return p; };
    for (int y = 0; y < L.h; y++) {
        uint8_t *row = inner + (size_t)y * out_stride;
        for (int x = 1; x <= border; x++) { row[-x] = row[refl(-x, L.w)]; row[L.w - 1 + x] = row[refl(L.w - 1 + x, L.w)]; }
    }
    for (int y = 1; y <= border; y++) {
        std::memcpy(inner + (long long)(-y) * out_stride - border, inner + (size_t)refl(-y, L.h) * out_stride - border, (size_t)L.w + 2 * border);
        std::memcpy(inner + (size_t)(L.h - 1 + y) * out_stride - border, inner + (size_t)refl(L.h - 1 + y, L.h) * out_stride - border, (size_t)L.w + 2 * border);
    }
    return RUMI_OK;
}

// Stage taps read the scratch arenas of the LAST chunk (device -> host on first use after a call).
static int fetch_taps(RumiOrb *h) {
    if (h->pending) { const int rc = rumi_orb_sync(h); if (rc != RUMI_OK) return rc; }
    if (h->tapValid) return RUMI_OK;
    const size_t nf = (size_t)h->lastChunkFrames;
    h->tapLevelStart.resize(nf * (kMaxLevels + 1));
    h->tapSelCount.resize(nf);
    h->tapCand.resize(nf * h->capCand);
    h->tapSelPacked.resize(nf * h->capSel);
    h->tapSelMeta.resize(nf * h->capSel);
    const size_t s0 = (size_t)h->lastChunkSlot;
    HIP_TRY(hipMemcpy(h->tapLevelStart.data(), h->dLevelStart + s0 * (kMaxLevels + 1), h->tapLevelStart.size() * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(h->tapSelCount.data(), h->dSelCount + s0, nf * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(h->tapCand.data(), h->dCand + s0 * h->capCand, h->tapCand.size() * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(h->tapSelPacked.data(), h->dSelPacked + s0 * h->capSel, h->tapSelPacked.size() * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(h->tapSelMeta.data(), h->dSelMeta + s0 * h->capSel, h->tapSelMeta.size() * 4, hipMemcpyDeviceToHost));
    h->tapValid = true;
    return RUMI_OK;
}

extern "C" int rumi_orb_stage_keypoints(RumiOrb *h, int32_t frame, int32_t level, int32_t stage, RumiKeyPoint *out,
                                        int32_t cap, int32_t *n_out) {
    if (!h || !n_out || h->lastFrames == 0 || level < 0 || level >= h->hP.nlevels) return RUMI_E_INVALID;
    const int f = frame - h->lastChunkBase;
    if (f < 0 || f >= h->lastChunkFrames) { g_lastError = "stage taps cover the frames of the last sub-chunk only"; return RUMI_E_INVALID; }
    HIP_TRY(hipSetDevice(h->device));
    int rc = fetch_taps(h);
    if (rc != RUMI_OK) return rc;
    if (stage == 0) {
        const int32_t *ls = h->tapLevelStart.data() + (size_t)f * (kMaxLevels + 1);
        const int n = ls[level + 1] - ls[level];
        *n_out = n;
        if (!out) return RUMI_OK;
        if (n > cap) return RUMI_E_CAPACITY;
        const uint32_t *c = h->tapCand.data() + (size_t)f * h->capCand + ls[level];
        for (int k = 0; k < n; k++)
            out[k] = RumiKeyPoint{(float)cand_x(c[k]), (float)cand_y(c[k]), 7.f, -1.f, (float)cand_score(c[k]), 0, -1};
        return RUMI_OK;
    }
    if (stage == 1) {
        const int tot = h->tapSelCount[f];
        const int ocap = h->lastOutCap;
        int n = 0;
        for (int k = 0; k < tot; k++) {
            const uint32_t meta = h->tapSelMeta[(size_t)f * h->capSel + k], pk = h->tapSelPacked[(size_t)f * h->capSel + k];
            if ((int)(meta & 0xFF) != level) continue;
            const int slot = (int)(meta >> 8);
            RumiKeyPoint kp;
            if (slot >= ocap) return RUMI_E_CAPACITY;
            HIP_TRY(hipMemcpy(&kp, (const uint8_t *)h->lastKp + (size_t)frame * h->lastKpStride + (size_t)slot * sizeof kp, sizeof kp, hipMemcpyDeviceToHost));
            kp.x = (float)(cand_x(pk) + kBorder); kp.y = (float)(cand_y(pk) + kBorder);
            if (out && n < cap) out[n] = kp;
            n++;
        }
        *n_out = n;
        return (out && n > cap) ? RUMI_E_CAPACITY : RUMI_OK;
    }
    return RUMI_E_INVALID;
}
