"""CPU tests of the oracle (test infrastructure): restated OpenCV primitives, constructor tables,
committed golden vectors.  The reference has no tests / golden vectors for this path, so
these pin the oracle against itself and against properties the primitives must have."""
import glob
import hashlib
import os

import numpy as np
import pytest

import oracle_lib as O
from rumi_slam_amd.synth import synth_frame

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_constructor_tables_match_survey():
    # values derived from ORBextractor.cc:405-461 for (1000, 1.2, 8)
    t = O.OracleExtractor(1000, 1.2, 8, 20, 7).tables()
    assert t["per_level"].tolist() == [217, 181, 151, 126, 105, 87, 73, 60]
    assert t["umax"].tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert O.OracleExtractor(2000, 1.2, 8, 20, 7).tables()["per_level"].tolist() == [434, 362, 302, 251, 209, 175, 145, 122]
    assert O.OracleExtractor(5000, 1.2, 8, 20, 7).tables()["per_level"].tolist() == [1086, 905, 754, 628, 524, 436, 364, 303]
    assert t["scale"][0] == 1.0 and abs(t["scale"][7] - 1.2 ** 7) < 1e-5


def test_level_sizes_match_survey():
    o = O.OracleExtractor()
    o.extract(synth_frame(1))
    sizes = [o.level(l).shape[::-1] for l in range(8)]
    assert sizes == [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]


def test_cv_round_half_even():
    L = O.lib()
    assert [L.orc_cv_round(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 2.5001)] == [0, 2, 2, 0, -2, 2, 3]


def test_fast_atan2_quadrants_and_accuracy():
    L = O.lib()
    assert L.orc_fast_atan2(0.0, 0.0) == 0.0
    rng = np.random.default_rng(0)
    for _ in range(2000):
        y, x = rng.normal(size=2) * 1000
        a = L.orc_fast_atan2(float(y), float(x))
        ref = np.degrees(np.arctan2(y, x)) % 360.0
        d = abs(a - ref)
        assert min(d, 360 - d) < 0.35        # cv documents ~0.3 deg
    assert abs(L.orc_fast_atan2(1.0, 0.0) - 90) < 0.01 and abs(L.orc_fast_atan2(0.0, -1.0) - 180) < 0.01


def test_resize_properties():
    flat = np.full((48, 64), 137, np.uint8)
    assert (O.resize_linear(flat, 53, 40) == 137).all()          # taps sum to one
    img = synth_frame(5)
    assert np.array_equal(O.resize_linear(img, 640, 480), img)    # identity scale -> fx = fy = 0
    half = O.resize_linear(img, 320, 240)                         # exact 2:1: mean of the 2x2 block, fixed point
    blk = img.reshape(240, 2, 320, 2).astype(np.int32)
    ref = (blk[:, 0, :, 0] + blk[:, 0, :, 1] + blk[:, 1, :, 0] + blk[:, 1, :, 1] + 2) >> 2
    assert np.abs(half.astype(np.int32) - ref).max() <= 1
    ramp = np.tile(np.arange(0, 240, dtype=np.uint8), (30, 1))
    r = O.resize_linear(ramp, 200, 25)
    assert (np.diff(r[3].astype(int)) >= 0).all()                 # monotone stays monotone


def test_gaussian_blur_properties():
    flat = np.full((40, 50), 201, np.uint8)
    assert (O.gaussian_blur(flat) == 201).all()                   # kernel sums to 256
    imp = np.zeros((31, 31), np.uint8)
    imp[15, 15] = 255
    b = O.gaussian_blur(imp).astype(np.int64)
    k = np.array([18, 34, 48, 56, 48, 34, 18], np.int64)
    ref = (np.outer(k, k) * 255 + 32768) >> 16
    assert np.array_equal(b[12:19, 12:19], ref)
    assert np.array_equal(b, b.T) and np.array_equal(b, b[::-1, ::-1])
    # REFLECT_101 border: blur of an image equals the centre of the blur of its reflect-padded version
    img = synth_frame(2)[:64, :80]
    pad = np.pad(img, 3, mode="reflect")
    assert np.array_equal(O.gaussian_blur(img), O.gaussian_blur(pad)[3:-3, 3:-3])


def test_fast_score_and_cell_on_synthetic_corner():
    img = np.full((21, 21), 100, np.uint8)
    img[10:, 10:] = 180          # an L-corner: pixel (10,10) sees a dark arc of 11 pixels
    img[10, 10] = 190            # break the score tie with (11,11) so strict-> NMS keeps exactly this one
    kps = O.fast_cell(img, 20)
    assert len(kps) == 1 and (kps[0]["x"], kps[0]["y"], kps[0]["response"]) == (10, 10, 89)
    assert all(3 <= k["x"] < 18 and 3 <= k["y"] < 18 for k in kps)
    # score definition: brute-force max-min over the 16 arcs of 9
    cx = [0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1]
    cy = [3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3]
    rng = np.random.default_rng(3)
    tile = rng.integers(0, 256, (9, 9), dtype=np.uint8)
    import ctypes as C
    c = np.ascontiguousarray(tile)
    got = O.lib().orc_fast_score(C.c_void_p(c.ctypes.data + 4 * 9 + 4), 9)
    d = [int(tile[4, 4]) - int(tile[4 + cy[k], 4 + cx[k]]) for k in range(16)]
    A = max(min(d[(k + j) % 16] for j in range(9)) for k in range(16))
    B = max(min(-d[(k + j) % 16] for j in range(9)) for k in range(16))
    assert got == max(A, B) - 1


def test_fast_nms_no_adjacent_and_threshold_monotone():
    img = synth_frame(9)[100:160, 200:270]
    k20, k7 = O.fast_cell(img, 20), O.fast_cell(img, 7)
    pts = {(int(k["x"]), int(k["y"])) for k in k20}
    for (x, y) in pts:
        assert not any((x + dx, y + dy) in pts for dx in (-1, 0, 1) for dy in (-1, 0, 1) if (dx, dy) != (0, 0))
    assert all(k["response"] >= 20 for k in k20) and all(k["response"] >= 7 for k in k7)
    assert [(k["y"], k["x"]) for k in k20] == sorted((k["y"], k["x"]) for k in k20)    # row-major emission


def test_quadtree_basic_properties():
    rng = np.random.default_rng(1)
    n = 3000
    pos = rng.choice(600 * 440, n, replace=False)
    cand = np.zeros(n, O.KP_DTYPE)
    cand["x"], cand["y"] = pos % 600, pos // 600
    cand["response"] = rng.integers(7, 255, n)
    out = O.octree(np.sort(cand, order=["y", "x"]), 16, 16 + 608, 16, 16 + 448, 217)
    assert 217 <= len(out) <= 217 + 3
    assert len({(k["x"], k["y"]) for k in out}) == len(out)
    few = O.octree(cand[:50], 16, 16 + 608, 16, 16 + 448, 217)
    assert len(few) == 50                                          # fewer candidates than wanted: all kept


def test_extract_empty_and_flat():
    o = O.OracleExtractor()
    mono, k, d = o.extract(np.full((480, 640), 90, np.uint8))
    assert mono == 0 and len(k) == 0


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "orb_*.npz"))))
def test_golden_vectors(path):
    g = np.load(path)
    w, h = g["wh"]
    kw = dict(n_rect=60, contrast=(8, 19)) if "lowtex" in path else {}
    img = synth_frame(int(g["seed"]), w=int(w), h=int(h), **kw)
    assert hashlib.sha256(img.tobytes()).hexdigest() == str(g["image_sha256"]), "synthetic generator changed"
    o = O.OracleExtractor(int(g["nfeatures"]), 1.2, 8, 20, 7)
    mono, kps, desc = o.extract(img, tuple(g["lap"]))
    assert mono == int(g["mono"])
    assert np.array_equal(kps.view(np.uint8).reshape(-1, 28), g["kps"])
    assert np.array_equal(desc, g["desc"])
    for l in range(8):
        assert hashlib.sha256(o.level(l).tobytes()).hexdigest() == str(g["level_sha256"][l])
