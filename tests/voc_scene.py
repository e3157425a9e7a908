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


def synthetic_vocabulary_fast(seed=0, k=10, L=6, stop_frac=0.05):
    """The regular tree of synthetic_vocabulary (ragged=False) built level by level with array operations: ORBvoc.txt's geometry -- k = 10, L = 6,
    1 111 111 nodes, 10^6 words -- in about a second (the node-by-node loop above needs minutes).  Same node model and file order (breadth-first,
    a node's children consecutive); the random stream differs from the loop's, so the two functions give different trees for one seed."""
    rng = np.random.default_rng(seed)
    parents, leafs, descs, weights = [np.array([-1], np.int32)], [np.zeros(1, np.uint8)], [np.zeros((1, 32), np.uint8)], [np.zeros(1)]
    first, prev_desc = 0, None                            # id of the first node of the previous level, its descriptors
    count = 1
    for level in range(1, L + 1):
        n_par = 1 if level == 1 else len(prev_desc)
        n = n_par * k
        par = np.repeat(np.arange(first, first + n_par, dtype=np.int32), k)
        d = rng.integers(0, 256, (n, 32), dtype=np.uint8) if level == 1 else np.repeat(prev_desc, k, axis=0)
        nflip = max(1, 24 >> level)
        rows = np.arange(n)
        for _ in range(nflip):                            # (a bit drawn twice flips back, as in the loop version)
            b = rng.integers(0, 256, n)
            d[rows, b >> 3] ^= (1 << (b & 7)).astype(np.uint8)
        is_leaf = level == L
        w = np.zeros(n)
        if is_leaf:
            w = rng.uniform(0.5, 9.0, n)
            w[rng.random(n) < stop_frac] = 0.0
        parents.append(par); leafs.append(np.full(n, 1 if is_leaf else 0, np.uint8)); descs.append(d); weights.append(w)
        first, prev_desc = count, d
        count += n
    return np.concatenate(parents), np.concatenate(leafs), np.concatenate(descs), np.concatenate(weights)


def write_text_fast(path, voc, k, L, scoring=0, weighting=0):
    """write_text for a million nodes in a few seconds: parent id, leaf flag and the 32 descriptor bytes are laid out as fixed-width decimal fields
    (leading zeros: operator>> and the library's parser read them as decimal) by array arithmetic; only the weights are formatted one by one."""
    parent, leaf, desc, weight = voc
    n = len(parent) - 1
    pw = 8
    fixed = np.full((n, pw + 1 + 2 + 32 * 4), ord(" "), np.uint8)
    pid = parent[1:].astype(np.int64)
    for j in range(pw):
        fixed[:, pw - 1 - j] = 48 + (pid // 10 ** j) % 10
    fixed[:, pw + 1] = 48 + leaf[1:]
    d = desc[1:].astype(np.int32)
    base = pw + 3
    fixed[:, base + 0:base + 128:4] = 48 + d // 100
    fixed[:, base + 1:base + 128:4] = 48 + (d // 10) % 10
    fixed[:, base + 2:base + 128:4] = 48 + d % 10
    head = fixed.view("S%d" % fixed.shape[1]).ravel()
    wtxt = np.char.mod("%.17g", weight[1:]).astype("S")
    lines = np.char.add(head, wtxt)
    with open(path, "wb") as f:
        f.write(f"{k} {L} {scoring} {weighting}\n".encode())
        f.write(b"\n".join(lines.tolist()))
        f.write(b"\n")
