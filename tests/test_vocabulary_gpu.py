"""GPU parity of the bag-of-words transform (DBoW2 tree descent + BowVector / FeatureVector assembly) against the CPU oracle:
word ids, node ids and the FeatureVector bit-exact; BowVector values bit-identical doubles (same additions in the same order)."""
import numpy as np
import pytest

import oracle_lib as O
from voc_scene import synthetic_vocabulary, synthetic_vocabulary_fast, write_text, write_text_fast

pytestmark = pytest.mark.gpu


def _features(voc, rng, n):
    """Descriptors near random leaves (plus pure noise), with duplicates so that words repeat inside one frame."""
    parent, leaf, desc, weight = voc
    leaves = np.nonzero(leaf)[0]
    pick = desc[rng.choice(leaves, n)].copy()
    for i in range(n):
        for b in rng.integers(0, 256, int(rng.integers(0, 30))):
            pick[i, b >> 3] ^= np.uint8(1 << (b & 7))
    pick[rng.random(n) < 0.1] = rng.integers(0, 256, (int((rng.random(n) < 0.1).sum()) or 1, 32), dtype=np.uint8)[0]
    return pick


@pytest.mark.parametrize("seed,k,L,ragged,levelsup", [(0, 10, 3, False, 1), (1, 10, 4, False, 2), (2, 7, 4, True, 2), (3, 16, 3, True, 4), (4, 20, 2, False, 0)])
def test_transform_matches_oracle(seed, k, L, ragged, levelsup):
    from rumi_slam_amd.vocabulary import ORBVocabulary
    voc = synthetic_vocabulary(seed, k, L, ragged)
    g, o = ORBVocabulary(*voc), O.OracleVocabulary(*voc)
    assert g.size() == int(voc[1].sum())
    rng = np.random.default_rng(100 + seed)
    d = _features(voc, rng, 1500)
    w1, v1, n1 = g.transform_features(d, levelsup)
    w2, v2, n2 = o.transform_features(d, levelsup)
    assert np.array_equal(w1, w2) and np.array_equal(n1, n2) and np.array_equal(v1, v2)
    (bi, bv), (fn, fo, fi) = g.transform(d, levelsup)
    (bi2, bv2), (fn2, fo2, fi2) = o.transform(d, levelsup)
    assert len(bi2) > 50 and len(bi2) < len(d), "words must repeat inside the frame for the sum order to matter"
    assert np.array_equal(bi, bi2) and bv.tobytes() == bv2.tobytes()
    assert np.array_equal(fn, fn2) and np.array_equal(fo, fo2) and np.array_equal(fi, fi2)
    assert abs(bv.sum() - 1.0) < 1e-12                                   # L1-normalised


@pytest.mark.parametrize("weighting,scoring", [(1, 0), (2, 1), (3, 5), (0, 5), (0, 1)])
def test_weighting_and_scoring_variants(weighting, scoring):
    from rumi_slam_amd.vocabulary import ORBVocabulary
    voc = synthetic_vocabulary(9, 8, 3)
    g, o = ORBVocabulary(*voc, weighting=weighting, scoring=scoring), O.OracleVocabulary(*voc, weighting=weighting, scoring=scoring)
    d = _features(voc, np.random.default_rng(5), 800)
    (bi, bv), fv = g.transform(d, 2)
    (bi2, bv2), fv2 = o.transform(d, 2)
    assert np.array_equal(bi, bi2) and bv.tobytes() == bv2.tobytes()
    for a, b in zip(fv, fv2):
        assert np.array_equal(a, b)


def test_text_file_loader_and_empty_input(tmp_path):
    from rumi_slam_amd.vocabulary import ORBVocabulary
    voc = synthetic_vocabulary(3, 5, 3)
    write_text(tmp_path / "voc.txt", voc, 5, 3)
    g, o = ORBVocabulary(path=tmp_path / "voc.txt"), O.OracleVocabulary(*voc)
    d = _features(voc, np.random.default_rng(1), 300)
    a, b = g.transform(d, 1), o.transform(d, 1)
    assert np.array_equal(a[0][0], b[0][0]) and a[0][1].tobytes() == b[0][1].tobytes()
    (bi, bv), (fn, fo, fi) = g.transform(np.zeros((0, 32), np.uint8))
    assert len(bi) == 0 and len(fn) == 0 and list(fo) == [0]


def test_header_levels_deeper_than_the_tree(tmp_path):
    """A text file whose header says L = 5 over a tree of depth 3: DBoW2 keeps the header's L for the FeatureVector level (nid_level = m_L - levelsup),
    so with levelsup = 3 the node ids are those of level 2, not of level 0."""
    from rumi_slam_amd.vocabulary import ORBVocabulary
    voc = synthetic_vocabulary(7, 6, 3)
    write_text(tmp_path / "voc.txt", voc, 6, 5)                 # header L = 5, deepest leaf at depth 3
    g, o = ORBVocabulary(path=tmp_path / "voc.txt"), O.OracleVocabulary(*voc)
    o.set_levels(5)
    d = _features(voc, np.random.default_rng(2), 400)
    for levelsup in (1, 3, 5, 6):
        w1, v1, n1 = g.transform_features(d, levelsup)
        w2, v2, n2 = o.transform_features(d, levelsup)
        assert np.array_equal(w1, w2) and np.array_equal(n1, n2), f"levelsup {levelsup}"
    o3 = O.OracleVocabulary(*voc)                               # the tree's own depth (3) would give other node ids at levelsup 3
    assert not np.array_equal(g.transform_features(d, 3)[2], o3.transform_features(d, 3)[2])


def test_batch_transform_feeds_search_by_bow():
    """Rumination batch: extract_batch -> transform_batch (all frames in one launch) == per-frame oracle transform; the
    resulting FeatureVectors drive SearchByBoW with the same matches as the oracle's."""
    import torch
    from rumi_slam_amd.extractor import ORBextractor
    from rumi_slam_amd.synth import synth_frame, warp_frame
    from rumi_slam_amd.vocabulary import ORBVocabulary
    voc = synthetic_vocabulary(11, 10, 3)
    g, o = ORBVocabulary(*voc), O.OracleVocabulary(*voc)
    img0 = synth_frame(31)
    img1, _ = warp_frame(img0, 32)
    frames = torch.from_numpy(np.stack([img0, img1, synth_frame(33)])).cuda()
    ext = ORBextractor(1000, 1.2, 8, 20, 7, max_batch=3)
    kp, desc, counts = ext.extract_batch(frames, (0, 1000), cap=1100)
    word, w, node = g.transform_batch(desc, counts, levelsup=2)
    torch.cuda.synchronize()
    for f in range(3):
        n = int(counts[f, 0])
        w2, v2, n2 = o.transform_features(desc[f, :n].cpu().numpy(), 2)
        assert np.array_equal(word[f, :n].cpu().numpy().view(np.uint32), w2)
        assert np.array_equal(node[f, :n].cpu().numpy().view(np.uint32), n2)
        assert np.array_equal(w[f, :n].cpu().numpy(), v2)


def _shuffle_levels(voc, seed):
    """The same tree with the node ids permuted inside every level (file order stays level order, parents still precede children): a node's children
    are no longer an id range, which is what routes the descent through the general kernel (child-list indirection)."""
    parent, leaf, desc, weight = voc
    n = len(parent)
    depth = np.zeros(n, np.int32)
    for i in range(1, n):
        depth[i] = depth[parent[i]] + 1
    rng = np.random.default_rng(seed)
    new_of_old = np.arange(n)
    for lv in range(1, depth.max() + 1):
        ids = np.nonzero(depth == lv)[0]
        new_of_old[ids] = rng.permutation(ids)
    old_of_new = np.argsort(new_of_old)
    p2 = np.where(parent[old_of_new] < 0, -1, new_of_old[np.maximum(parent[old_of_new], 0)]).astype(np.int32)
    return p2, leaf[old_of_new].copy(), desc[old_of_new].copy(), weight[old_of_new].copy()


def test_general_tree_whose_children_are_not_id_ranges():
    from rumi_slam_amd.vocabulary import ORBVocabulary
    voc = _shuffle_levels(synthetic_vocabulary(5, 9, 4, True), 1)
    kids = {}
    for i in range(1, len(voc[0])):
        kids.setdefault(int(voc[0][i]), []).append(i)
    assert any(np.any(np.diff(v) != 1) for v in kids.values()), "the permutation must break the id ranges"
    g, o = ORBVocabulary(*voc), O.OracleVocabulary(*voc)
    d = _features(voc, np.random.default_rng(3), 1200)
    for levelsup in (1, 2, 4):
        a, b = g.transform_features(d, levelsup), o.transform_features(d, levelsup)
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), f"levelsup {levelsup}"
    a, b = g.transform(d, 2), o.transform(d, 2)
    assert np.array_equal(a[0][0], b[0][0]) and a[0][1].tobytes() == b[0][1].tobytes() and all(np.array_equal(x, y) for x, y in zip(a[1], b[1]))


def test_orbvoc_geometry_k10_L6_levelsup4(tmp_path):
    """The reference's real vocabulary geometry: Frame::ComputeBoW calls transform(..., 4) (Frame.cc:763-768) on ORBvoc.txt's k = 10, L = 6 tree
    (1 111 111 nodes, 10^6 words; the file itself is a missing blob: a synthetic tree of the same shape, built in memory AND written to / read back
    from a 170 MB text file by rumi_voc_load_text).  2000 features: word, weight, node per feature and the assembled BowVector / FeatureVector
    identical to the oracle's; the FeatureVector groups by level-2 nodes (100 of them), as levelsup = 4 means on a 6-level tree."""
    import time
    from rumi_slam_amd.vocabulary import ORBVocabulary
    voc = synthetic_vocabulary_fast(3, 10, 6)
    assert len(voc[0]) == 1111111 and int(voc[1].sum()) == 10 ** 6
    o = O.OracleVocabulary(*voc)
    rng = np.random.default_rng(8)
    d = _features(voc, rng, 2000)
    t0 = time.time(); write_text_fast(tmp_path / "big.txt", voc, 10, 6); t_write = time.time() - t0
    t0 = time.time(); gf = ORBVocabulary(path=tmp_path / "big.txt"); t_load = time.time() - t0
    gm = ORBVocabulary(*voc)
    print(f"ORBvoc-sized tree: text file written in {t_write:.1f} s, rumi_voc_load_text {t_load:.2f} s")
    assert gf.size() == 10 ** 6 and gf.levels() == 6 and gm.levels() == 6
    for g in (gf, gm):
        for levelsup in (4, 2, 6):
            a, b = g.transform_features(d, levelsup), o.transform_features(d, levelsup)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), f"levelsup {levelsup}"
        (bi, bv), (fn, fo, fi) = g.transform(d, 4)
        (bi2, bv2), (fn2, fo2, fi2) = o.transform(d, 4)
        assert np.array_equal(bi, bi2) and bv.tobytes() == bv2.tobytes()
        assert np.array_equal(fn, fn2) and np.array_equal(fo, fo2) and np.array_equal(fi, fi2)
        assert 20 < len(fn) <= 110 and fn.min() >= 11 and fn.max() <= 110          # level-2 node ids: 11 .. 110
    t0 = time.time()
    for _ in range(20):
        gm.transform_features(d, 4)
    print(f"transform_features, 2000 features, host arrays in and out: {(time.time() - t0) / 20 * 1e3:.3f} ms per call")
