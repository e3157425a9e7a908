"""CPU checks of the vocabulary oracle (no GPU): the restated DBoW2 descent against an independent numpy arg-min walk, and the
BowVector / FeatureVector assembly rules (sum in feature order, stopped words dropped, L1 norm)."""
import numpy as np

import oracle_lib as O
from voc_scene import synthetic_vocabulary

POP = np.array([bin(i).count("1") for i in range(256)], np.int32)


def _walk(voc, d, levelsup, L):
    parent, leaf, desc, weight = voc
    children = {}
    for i in range(1, len(parent)):
        children.setdefault(int(parent[i]), []).append(i)
    word_of = {}
    for i in range(1, len(parent)):
        if leaf[i]:
            word_of[i] = len(word_of)
    node, level, nid = 0, 0, 0
    while node in children:
        ch = children[node]
        dist = [int(POP[desc[c] ^ d].sum()) for c in ch]
        node = ch[int(np.argmin(dist))]            # argmin takes the first minimum, like the strict `<` of the reference
        level += 1
        if level == L - levelsup:
            nid = node
    return word_of[node], weight[node], nid


def test_descent_equals_independent_walk():
    voc = synthetic_vocabulary(5, 6, 3, ragged=True)
    o = O.OracleVocabulary(*voc)
    rng = np.random.default_rng(0)
    d = rng.integers(0, 256, (200, 32), dtype=np.uint8)
    parent, leaf = voc[0], voc[1]
    depth = np.zeros(len(parent), int)
    for i in range(1, len(parent)):
        depth[i] = depth[parent[i]] + 1
    L = int(depth[leaf == 1].max())
    w, v, n = o.transform_features(d, 1)
    for i in range(len(d)):
        ww, vv, nn = _walk(voc, d[i], 1, L)
        assert (w[i], v[i], n[i]) == (ww, vv, nn)


def test_bow_assembly_rules():
    voc = synthetic_vocabulary(6, 4, 2, stop_frac=0.3)
    o = O.OracleVocabulary(*voc)
    rng = np.random.default_rng(1)
    d = rng.integers(0, 256, (400, 32), dtype=np.uint8)
    w, v, n = o.transform_features(d, 1)
    (bi, bv), (fn, fo, fi) = o.transform(d, 1)
    live = v > 0
    assert not live.all() and live.any()
    assert np.array_equal(bi, np.unique(w[live]))
    sums = {}
    for i in np.nonzero(live)[0]:                      # feature order
        sums[int(w[i])] = sums.get(int(w[i]), 0.0) + float(v[i])
    norm = 0.0
    for k in sorted(sums):
        norm += abs(sums[k])
    assert np.array_equal(bv, np.array([sums[int(k)] / norm for k in bi]))
    assert np.array_equal(fn, np.unique(n[live]))
    for a, node in enumerate(fn):
        assert np.array_equal(fi[fo[a]:fo[a + 1]], np.nonzero(live & (n == node))[0])
