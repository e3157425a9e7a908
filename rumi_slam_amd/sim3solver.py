"""Host-side mirror of ``ORB_SLAM3::Sim3Solver`` (R/include/cloud_edge_slam_lib/Sim3Solver.h, R/lib_src/Sim3Solver.cc) over
``rumi_sim3_ransac`` (include/rumi_opt.h): the RANSAC state machine stays on the host, every block of iterations is one launch.

The minimal sets of upstream come from ``DUtils::Random::RandomInt`` = glibc ``rand()`` (Thirdparty/DBoW2/DUtils/Random.cpp:47-50).  The draws
do not depend on the hypotheses' results, so a whole block is drawn first, evaluated on the GPU (one workgroup per hypothesis) and the
"best so far / converged" logic is replayed over the results in iteration order.  ``GlibcRand`` restates glibc's TYPE_3 generator so
that a block that converges early leaves the generator exactly where upstream's loop would have left it (tests pin it against the
real ``srand`` / ``rand`` of this image)."""
import math

import numpy as np


class GlibcRand:
    """glibc ``srand(seed)`` / ``rand()`` (random_r.c, TYPE_3: x[i] = x[i-3] + x[i-31], output >> 1)."""

    def __init__(self, seed=1):
        self.seed(seed)

    def seed(self, seed):
        seed = int(seed) & 0xFFFFFFFF
        if seed == 0:
            seed = 1
        r = [0] * 34
        r[0] = seed
        for i in range(1, 31):
            # 16807 * r[i-1] % 2147483647 on the signed 32-bit word, as __srandom_r computes it
            word = r[i - 1] if r[i - 1] < 0x80000000 else r[i - 1] - (1 << 32)
            hi, lo = int(word / 127773), int(math.fmod(word, 127773))
            word = 16807 * lo - 2836 * hi
            if word < 0:
                word += 2147483647
            r[i] = word
        for i in range(31, 34):
            r[i] = r[i - 31]
        self._r = [x & 0xFFFFFFFF for x in r]
        for _ in range(310):
            self._next_word()

    def _next_word(self):
        r = self._r
        v = (r[-31] + r[-3]) & 0xFFFFFFFF
        r.append(v)
        del r[0]
        return v

    def rand(self):
        return self._next_word() >> 1

    def state(self):
        return list(self._r)

    def set_state(self, st):
        self._r = list(st)

    def RandomInt(self, lo, hi):                       # DUtils/Random.cpp:47-50
        d = hi - lo + 1
        return int((float(self.rand()) / (2147483647.0 + 1.0)) * d) + lo


class Sim3Solver:
    """Sim3Solver(pKF1, pKF2, vpMatched12, bFixScale, vpKeyFrameMatchedMP) on flat arrays: one entry per correspondence the constructor
    keeps (:78-126).  X3Dc1 / X3Dc2 = mvX3Dc1 / mvX3Dc2, sigma2_1 / sigma2_2 = mvLevelSigma2 at the two key-points' octaves,
    indices1 = mvnIndices1 (position of every kept correspondence in vpMatched12, length mN1)."""

    def __init__(self, optimizer, X3Dc1, X3Dc2, sigma2_1, sigma2_2, K4_1, K4_2, fix_scale=False, indices1=None, mN1=None, rng=None):
        self._opt = optimizer
        self.X1 = np.ascontiguousarray(X3Dc1, np.float32).reshape(-1, 3)
        self.X2 = np.ascontiguousarray(X3Dc2, np.float32).reshape(-1, 3)
        self.s1, self.s2 = np.ascontiguousarray(sigma2_1, np.float32), np.ascontiguousarray(sigma2_2, np.float32)
        self.K1, self.K2 = np.ascontiguousarray(K4_1, np.float32), np.ascontiguousarray(K4_2, np.float32)
        self.fix_scale = bool(fix_scale)
        self.N = len(self.X1)
        self.indices1 = np.arange(self.N) if indices1 is None else np.asarray(indices1, np.int64)
        self.mN1 = int(mN1) if mN1 is not None else (int(self.indices1.max()) + 1 if self.N else 0)
        self.rng = rng if rng is not None else GlibcRand(0)      # DUtils::Random::SeedRandOnce(0) of upstream's solvers
        self.mnIterations = 0
        self.mnBestInliers = 0
        self.mvbBestInliers = np.zeros(self.N, bool)
        self.mBestRotation, self.mBestTranslation, self.mBestScale = np.eye(3, dtype=np.float32), np.zeros(3, np.float32), np.float32(1)
        self._last = None                                        # transform of the previous iteration (kept on a degenerate set, :493-494)
        self.SetRansacParameters()

    def SetRansacParameters(self, probability=0.99, minInliers=6, maxIterations=300):      # :134-157
        self.mRansacProb, self.mRansacMinInliers = float(probability), int(minInliers)
        N = self.N
        if self.mRansacMinInliers == N:
            n_it = 1
        else:
            eps = float(np.float32(self.mRansacMinInliers) / np.float32(N)) if N else 0.0
            den = math.log(1 - eps ** 3) if 0 < eps < 1 else 0.0
            n_it = int(math.ceil(math.log(1 - self.mRansacProb) / den)) if den != 0 else maxIterations
        self.mRansacMaxIts = max(1, min(n_it, int(maxIterations)))
        self.mnIterations = 0

    # ---- one block of iterations -----------------------------------------------------------------------------------------------
    def _draw_block(self, count):
        """count x 3 indices exactly as :176-191, plus the generator state after every hypothesis (to stop where upstream's loop stops)."""
        tri, states = np.zeros((count, 3), np.int32), []
        for h in range(count):
            avail = list(range(self.N))
            for i in range(3):
                r = self.rng.RandomInt(0, len(avail) - 1)
                tri[h, i] = avail[r]
                avail[r] = avail[-1]
                avail.pop()
            states.append(self.rng.state())
        return tri, states

    def _block(self, nIterations, score=None):
        count = max(0, min(int(nIterations), self.mRansacMaxIts - self.mnIterations))
        if count == 0:
            return None, None, 0
        tri, states = self._draw_block(count)
        res = self._opt.Sim3Ransac(self.X1, self.X2, self.s1, self.s2, self.K1, self.K2, tri, fix_scale=self.fix_scale, score=score)
        return res, states, count

    def _hyp(self, res, h):
        """Result of iteration h; a degenerate minimal set keeps the previous iteration's transform and inliers."""
        if res["valid"][h] or self._last is None:
            cur = dict(R=res["R"][h], t=res["t"][h], s=res["s"][h], n=int(res["n_inliers"][h]), inl=res["inliers"][h],
                       ratio=float(res["median"][h]) if res["median"] is not None else None)
            if not res["valid"][h]:
                cur["n"], cur["inl"] = 0, np.zeros(self.N, bool)
            self._last = cur
        return self._last

    def _take_best(self, cur):
        self.mvbBestInliers, self.mnBestInliers = cur["inl"].copy(), cur["n"]
        self.mBestRotation, self.mBestTranslation, self.mBestScale = cur["R"].copy(), cur["t"].copy(), np.float32(cur["s"])

    def _T12(self):
        T = np.eye(4, dtype=np.float32)
        T[:3, :3] = self.mBestScale * self.mBestRotation
        T[:3, 3] = self.mBestTranslation
        return T

    def _vb(self, inl):
        vb = np.zeros(self.mN1, bool)
        vb[self.indices1[inl]] = True
        return vb

    def iterate(self, nIterations):
        """:159-220.  Returns (T12 or identity, bNoMore, vbInliers, nInliers)."""
        vb = np.zeros(self.mN1, bool)
        if self.N < self.mRansacMinInliers:
            return np.eye(4, dtype=np.float32), True, vb, 0
        res, states, count = self._block(nIterations)
        for h in range(count):
            self.mnIterations += 1
            cur = self._hyp(res, h)
            if cur["n"] >= self.mnBestInliers:
                self._take_best(cur)
                if cur["n"] > self.mRansacMinInliers:
                    self.rng.set_state(states[h])
                    return self._T12(), False, self._vb(cur["inl"]), cur["n"]
        return np.eye(4, dtype=np.float32), self.mnIterations >= self.mRansacMaxIts, vb, 0

    def iterate_converge(self, nIterations):
        """:222-290.  Returns (best T12 of this call or None, bNoMore, vbInliers, nInliers, bConverge)."""
        vb = np.zeros(self.mN1, bool)
        if self.N < self.mRansacMinInliers:
            return np.eye(4, dtype=np.float32), True, vb, 0, False
        res, states, count = self._block(nIterations)
        best, n_in = None, 0
        for h in range(count):
            self.mnIterations += 1
            cur = self._hyp(res, h)
            if cur["n"] >= self.mnBestInliers:
                self._take_best(cur)
                n_in = cur["n"]
                if cur["n"] > self.mRansacMinInliers:
                    self.rng.set_state(states[h])
                    return self._T12(), False, self._vb(cur["inl"]), n_in, True
                best = self._T12()
        return best, self.mnIterations >= self.mRansacMaxIts, vb, n_in, False

    def iterate_rumination(self, nIterations, score, bestRatio):
        """The overload of the sub-map merge, :292-404 (CloudMerging.cc:717): every hypothesis is also scored with ComputeInliersNum over all
        key-frame pairs.  Returns (best T12 of this call or None, bNoMore, bConverge, bestRatio); the best rotation / translation / scale
        are the members mBestRotation / mBestTranslation / mBestScale (upstream's bestRotation, bestTranslation, bestScale outputs)."""
        if self.N < self.mRansacMinInliers:
            return np.eye(4, dtype=np.float32), True, False, bestRatio
        res, states, count = self._block(nIterations, score)
        best = None
        for h in range(count):
            self.mnIterations += 1
            cur = self._hyp(res, h)
            if cur["ratio"] >= bestRatio and cur["n"] >= self.mnBestInliers:
                self._take_best(cur)
                bestRatio = cur["ratio"]
                if cur["ratio"] > 0.10 and cur["n"] > self.mRansacMinInliers:
                    self.rng.set_state(states[h])
                    return self._T12(), False, True, bestRatio
                best = self._T12()
        return best, self.mnIterations >= self.mRansacMaxIts, False, bestRatio

    def find(self):                                    # :425-428
        T, _, vb, n = self.iterate(self.mRansacMaxIts)
        return T, vb, n

    def GetEstimatedRotation(self):
        return self.mBestRotation

    def GetEstimatedTranslation(self):
        return self.mBestTranslation

    def GetEstimatedScale(self):
        return float(self.mBestScale)
