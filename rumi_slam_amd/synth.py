"""Seeded synthetic inputs for the hot path.  Integer arithmetic only, so the
same seed gives the same bytes on the build container and on the GPU box.

Frames: 640x480 u8 — smooth background (6 low-frequency sinusoids, total amplitude 40, mean 110)
+ ``n_rect`` random axis-aligned / rotated solid or checker rectangles with contrast U[25,120]
+ i.i.d. noise U{-3..3}, clamped to [0,255].  ``n_rect=0`` gives the low-texture set that
exercises the minThFAST retry of the extractor.
"""
import numpy as np

_LUT_N = 1024
# sin LUT, scale 2**14 (rounded once; ulp-level libm differences cannot change a rounded entry)
_SIN = np.round(np.sin(np.arange(_LUT_N) * (2.0 * np.pi / _LUT_N)) * 16384.0).astype(np.int64)


def synth_frame(seed, w=640, h=480, n_rect=400, noise=3, contrast=(25, 120)):
    rng = np.random.Generator(np.random.PCG64(int(seed)))
    yy, xx = np.mgrid[0:h, 0:w].astype(np.int64)
    acc = np.full((h, w), 110 << 14, dtype=np.int64)
    amps = [10, 8, 7, 6, 5, 4]  # sums to 40
    for a in amps:
        fx, fy = int(rng.integers(0, 4)), int(rng.integers(0, 4))
        if fx == 0 and fy == 0:
            fx = 1
        ph = int(rng.integers(0, _LUT_N))
        idx = (xx * (fx * _LUT_N) // w + yy * (fy * _LUT_N) // h + ph) & (_LUT_N - 1)
        acc += a * _SIN[idx]
    img = acc >> 14
    for _ in range(n_rect):
        cx, cy = int(rng.integers(0, w)), int(rng.integers(0, h))
        hw, hh = int(rng.integers(4, 41)), int(rng.integers(4, 41))
        rot = int(rng.integers(0, 2))
        ang = int(rng.integers(0, _LUT_N)) if rot else 0
        checker = int(rng.integers(0, 4)) == 0
        cs = int(rng.integers(4, 9))
        con = int(rng.integers(contrast[0], contrast[1] + 1)) * (1 if int(rng.integers(0, 2)) else -1)
        r = int(1.5 * max(hw, hh)) + 2
        x0, x1, y0, y1 = max(cx - r, 0), min(cx + r + 1, w), max(cy - r, 0), min(cy + r + 1, h)
        if x0 >= x1 or y0 >= y1:
            continue
        dx = xx[y0:y1, x0:x1] - cx
        dy = yy[y0:y1, x0:x1] - cy
        c, s = int(_SIN[(ang + _LUT_N // 4) & (_LUT_N - 1)]), int(_SIN[ang])
        u = (dx * c + dy * s) >> 14
        v = (dy * c - dx * s) >> 14
        m = (np.abs(u) < hw) & (np.abs(v) < hh)
        if checker:
            sign = 1 - 2 * ((((u + 64 * cs) // cs) + ((v + 64 * cs) // cs)) & 1)
            img[y0:y1, x0:x1] += np.where(m, con * sign, 0)
        else:
            img[y0:y1, x0:x1] += np.where(m, con, 0)
    if noise:
        img = img + rng.integers(-noise, noise + 1, size=(h, w))
    return np.clip(img, 0, 255).astype(np.uint8)


def synth_batch(n, seed0=1234, **kw):
    """Frames seed0, seed0+1, ... as one (n, h, w) u8 array."""
    return np.stack([synth_frame(seed0 + i, **kw) for i in range(n)])


def warp_frame(img, seed, max_rot_deg=3.0, max_shift=8, max_scale=0.02):
    """Frame t+1 = frame t under a small similarity warp; bilinear in
    16.16 fixed point, edge-clamped.  Returns (warped, A) with A the 2x3 forward map (float64)."""
    h, w = img.shape
    rng = np.random.Generator(np.random.PCG64(int(seed)))
    rot = (rng.random() * 2 - 1) * max_rot_deg * np.pi / 180.0
    sc = 1.0 + (rng.random() * 2 - 1) * max_scale
    tx, ty = (rng.random(2) * 2 - 1) * max_shift
    ca, sa = np.cos(rot) * sc, np.sin(rot) * sc
    cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
    # forward map p' = R(p-c)+c+t ; sample source at inverse
    A = np.array([[ca, -sa, cx + tx - ca * cx + sa * cy], [sa, ca, cy + ty - sa * cx - ca * cy]])
    Ai = np.linalg.inv(np.vstack([A, [0, 0, 1]]))[:2]
    Q = np.round(Ai * 65536.0).astype(np.int64)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.int64)
    sx = Q[0, 0] * xx + Q[0, 1] * yy + Q[0, 2]
    sy = Q[1, 0] * xx + Q[1, 1] * yy + Q[1, 2]
    ix, iy = sx >> 16, sy >> 16
    fx, fy = sx & 65535, sy & 65535
    ix0, ix1 = np.clip(ix, 0, w - 1), np.clip(ix + 1, 0, w - 1)
    iy0, iy1 = np.clip(iy, 0, h - 1), np.clip(iy + 1, 0, h - 1)
    im = img.astype(np.int64)
    top = im[iy0, ix0] * (65536 - fx) + im[iy0, ix1] * fx
    bot = im[iy1, ix0] * (65536 - fx) + im[iy1, ix1] * fx
    out = (top * (65536 - fy) + bot * fy + (1 << 31)) >> 32
    return np.clip(out, 0, 255).astype(np.uint8), A


def warp_homography(img, H):
    """Image of a plane under a camera motion: out(x') = img(H^-1 x'), bilinear in float64, edge-clamped.  H maps source
    pixels to destination pixels (3x3).  Used by the closed-loop tracking substitute for BASELINE config 1."""
    h, w = img.shape
    Hi = np.linalg.inv(np.asarray(H, np.float64))
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    den = Hi[2, 0] * xx + Hi[2, 1] * yy + Hi[2, 2]
    sx = (Hi[0, 0] * xx + Hi[0, 1] * yy + Hi[0, 2]) / den
    sy = (Hi[1, 0] * xx + Hi[1, 1] * yy + Hi[1, 2]) / den
    ix, iy = np.floor(sx).astype(np.int64), np.floor(sy).astype(np.int64)
    fx, fy = sx - ix, sy - iy
    ix0, ix1 = np.clip(ix, 0, w - 1), np.clip(ix + 1, 0, w - 1)
    iy0, iy1 = np.clip(iy, 0, h - 1), np.clip(iy + 1, 0, h - 1)
    im = img.astype(np.float64)
    out = (im[iy0, ix0] * (1 - fx) + im[iy0, ix1] * fx) * (1 - fy) + (im[iy1, ix0] * (1 - fx) + im[iy1, ix1] * fx) * fy
    return np.clip(np.floor(out + 0.5), 0, 255).astype(np.uint8)
