"""Synthetic vocabulary trees with the node model of DBoW2's text format (ORBvoc.txt itself is a missing blob)."""
import numpy as np


def synthetic_vocabulary(seed=0, k=10, L=3, ragged=False, stop_frac=0.05):
    """Node table (parent, is_leaf, desc, weight) in file order: breadth-first, children of a node consecutive.
    Child descriptors are noisy copies of their parent's, so descents are meaningful; weights are idf-like, a few words
    are stopped (weight 0).  ragged=True varies the branching factor (1..k) and makes some leaves shallow."""
    rng = np.random.default_rng(seed)
    parent, leaf, desc, weight, depth = [-1], [0], [np.zeros(32, np.uint8)], [0.0], [0]
    frontier = [0]
    for level in range(1, L + 1):
        nxt = []
        for p in frontier:
            nchild = int(rng.integers(1, k + 1)) if ragged else k
            for _ in range(nchild):
                d = desc[p].copy() if p else rng.integers(0, 256, 32, dtype=np.uint8)
                flip = rng.integers(0, 256, max(1, 24 >> level))
                for b in flip:
                    d[b >> 3] ^= np.uint8(1 << (b & 7))
                nid = len(parent)
                is_leaf = level == L or (ragged and level >= 2 and rng.random() < 0.15)
                parent.append(p); leaf.append(int(is_leaf)); desc.append(d); depth.append(level)
                weight.append(0.0 if (is_leaf and rng.random() < stop_frac) else float(rng.uniform(0.5, 9.0)) if is_leaf else 0.0)
                if not is_leaf:
                    nxt.append(nid)
        frontier = nxt
    return (np.array(parent, np.int32), np.array(leaf, np.uint8), np.stack(desc).astype(np.uint8), np.array(weight, np.float64))


def write_text(path, voc, k, L, scoring=0, weighting=0):
    parent, leaf, desc, weight = voc
    with open(path, "w") as f:
        f.write(f"{k} {L} {scoring} {weighting}\n")
        for i in range(1, len(parent)):
            f.write(f"{parent[i]} {leaf[i]} " + " ".join(str(int(b)) for b in desc[i]) + f" {float(weight[i])!r}\n")
