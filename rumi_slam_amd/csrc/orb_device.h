// Device-visible parameter blocks of the ORB extractor and the kernel launch wrappers.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "orb_geom.h"
#include "rumi_orb.h"

namespace rumi {

struct DevLevel {
    int w, h, pitch;
    long long off;                 // byte offset of the level in a frame's arena (pyramid and blur arenas alike)
    int nCols, nRows, wCell, hCell;
    int cellBase, nCells;
    int maxBX, maxBY;
    int nfeat;
    float scale;                   // mvScaleFactor[level]
    float patchSize;               // (float)(int)(31 * scale)  (ORBextractor.cc:816)
    int candCap;
    int coefX, coefXT, coefY, xmax; // resize tables (int16 units into the coefficient buffer): column offsets, column tap pairs, row tables;
                                   // the column tables carry 4 more entries than the level has columns (copies of the last one), so that the
                                   // lane holding the row's last, partial dword computes 4 outputs like every other lane
    int xmaxFast;                  // xmax, or w + 3 when no column clamps its second tap (then every lane's 4 outputs may take the fast path)
    int rowTab;                    // first entry of the level's output-row table (RowTap units, row 0 first)
};

struct DevParams {
    int nlevels, totalCells, maxCellCand, totalCand;
    int iniTh, minTh;
    long long arenaStride;         // bytes per frame in the pyramid / blur arenas
    int umax[16];
    DevLevel lv[kMaxLevels];
};

// Where the kernels find pixels: level 0 IS the caller's frame (4-byte aligned base / pitch / frame stride; the host stages a frame
// that is not), levels 1.. live in `pyr`, the blurred levels (0 included) in `blur` with the same per-level offsets.  No borders anywhere.
struct ImgSrc {
    const uint8_t *l0;             // caller's frames = level 0
    long long l0FrameStride;
    int l0Pitch;
    uint8_t *pyr;
    uint8_t *blur;
};

struct RowTap { int32_t r0, r1; uint32_t bh0, bh1; };   // the two (clamped) source rows, vertical taps << 16
// The whole pyramid in ONE launch for calls of a few frames (k_pyramid_tiles): per workgroup and level the region it computes (multiples of 4 in x;
// level 0: the window of the caller's frame it reads), derived on the host from the resize tables
struct PyrTile { int16_t x0[kMaxLevels], x1[kMaxLevels], y0[kMaxLevels], y1[kMaxLevels]; };
void launch_pyramid_tiles(const DevParams *dP, ImgSrc src, const int16_t *coef, const RowTap *rowTab, const PyrTile *tiles, int ntiles, int bufBytes,
                          int tabEntries, int nframes, hipStream_t st, int32_t *clearWord = nullptr, bool copyL0 = false);
void launch_resize(const DevParams *dP, const DevParams &hP, ImgSrc src, const int16_t *coef, const RowTap *rowTab, int level, int nframes,
                   hipStream_t st, int32_t *clearWord = nullptr);
void launch_fast(const DevParams *dP, const DevParams &hP, ImgSrc src, uint32_t *cellBuf, int32_t *cellCnt, int nframes,
                 hipStream_t st);
bool fast_blur_fusable(const DevParams &hP);
bool launch_fast_blur(const DevParams *dP, const DevParams &hP, ImgSrc src, uint32_t *cellBuf, int32_t *cellCnt, int nframes, int variant, hipStream_t st);
void launch_compact(const DevParams *dP, const DevParams &hP, const uint32_t *cellBuf, const int32_t *cellCnt,
                    uint32_t *cand, int32_t *levelStart, int32_t *errFlag, int nframes, hipStream_t st);
void launch_blur(const DevParams *dP, const DevParams &hP, ImgSrc src, int nframes, int variant, hipStream_t st);
void launch_orient_desc(const DevParams *dP, ImgSrc src, const uint32_t *selPacked, const uint32_t *selMeta,
                        const int32_t *selCount, int selCap, int maxSel, RumiKeyPoint *kpOut, long long kpStride, uint8_t *descOut,
                        long long descStride, int outCap, int nframes, hipStream_t st);   // strides: bytes from one frame's outputs to the next

void launch_assemble_orient_desc(const DevParams *dP, ImgSrc src, const uint32_t *selLevel, const int32_t *selLevelCnt, int selLevelCap, int lap0, int lap1,
                                 int32_t *counts, long long countsStride, int32_t *errFlag, int32_t *errMirror, uint32_t *selPacked, uint32_t *selMeta,
                                 int32_t *selCount, int selCap, int maxSel, RumiKeyPoint *kpOut,
                                 long long kpStride, uint8_t *descOut, long long descStride, int outCap, int nframes, hipStream_t st);
void launch_octree(const DevParams *dP, const DevParams &hP, const uint32_t *cand, const int32_t *levelStart,
                   uint16_t *owner, uint32_t *selLevel, int32_t *selLevelCnt, int selLevelCap, int32_t *errFlag,
                   int nframes, size_t ldsBytes, hipStream_t st);
void launch_assemble(const DevParams *dP, const uint32_t *selLevel, const int32_t *selLevelCnt, int selLevelCap, int lap0,
                     int lap1, uint32_t *selPacked, uint32_t *selMeta, int32_t *selCount, int selCap, int32_t *counts, long long countsStride,
                     int32_t *errFlag, int nframes, hipStream_t st, int32_t *errMirror = nullptr);
size_t octree_lds_for(const DevParams &hP);

}  // namespace rumi
