"""GPU parity of the Hamming matchers against the CPU oracle (match indices bit-exact), through the C ABI."""
import numpy as np
import pytest

import oracle_lib as O
from scene import K_TUM3, TrackingScene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def matcher():
    from rumi_slam_amd.matcher import ORBmatcher
    return lambda nn=0.6, ori=True: ORBmatcher(nn, ori)


def _fv(mapping):
    from rumi_slam_amd.matcher import FeatureVector
    return FeatureVector(mapping)


@pytest.mark.parametrize("seed,th", [(0, 15.0), (1, 30.0), (2, 7.0), (3, 15.0)])
def test_search_by_projection_frame(matcher, seed, th):
    from rumi_slam_amd.matcher import FrameView
    s = TrackingScene(seed)
    cur = FrameView(s.cur_keys, s.cur_desc, s.w, s.h, s.sf)
    cur_mp0 = np.full(cur.n, -1, np.int32)
    for ori in (True, False):
        n_ref, ref = O.search_by_projection_frame(s.cur_keys, s.cur_desc, s.w, s.h, s.sf, s.Tcw7, K_TUM3, s.last_keys, s.last_mp,
                                                  s.last_outlier, s.mp_pos, s.mp_desc, s.mp_obs, cur_mp0, th, ori)
        n_gpu, got = matcher(0.9, ori).SearchByProjection_Frame(cur, s.Tcw7, K_TUM3, s.last_keys, s.last_mp, s.last_outlier, s.mp_pos,
                                                               s.mp_desc, s.mp_obs, cur_mp0, th)
        assert n_ref > 100, "scene should produce matches"
        assert n_gpu == n_ref
        assert np.array_equal(got, ref), f"{np.count_nonzero(got != ref)} features differ"


def test_candidate_lists_larger_than_a_slot():
    """The one-launch candidate pass gives every query a fixed slot of the list arena; a radius that makes most of the frame a candidate of every
    query does not fit, and the search must fall back to count / scan / fill (and grow the arena) with the same result."""
    from rumi_slam_amd.matcher import FrameView, ORBmatcher
    s = TrackingScene(3)
    cur = FrameView(s.cur_keys, s.cur_desc, s.w, s.h, s.sf)
    cur_mp0 = np.full(cur.n, -1, np.int32)
    th = 400.0
    n_ref, ref = O.search_by_projection_frame(s.cur_keys, s.cur_desc, s.w, s.h, s.sf, s.Tcw7, K_TUM3, s.last_keys, s.last_mp, s.last_outlier, s.mp_pos,
                                              s.mp_desc, s.mp_obs, cur_mp0, th, True)
    m = ORBmatcher(0.9, True, max_features=2048, max_queries=2048)
    n_gpu, got = m.SearchByProjection_Frame(cur, s.Tcw7, K_TUM3, s.last_keys, s.last_mp, s.last_outlier, s.mp_pos, s.mp_desc, s.mp_obs, cur_mp0, th)
    assert n_gpu == n_ref and np.array_equal(got, ref), f"{np.count_nonzero(got != ref)} features differ"
    n_gpu, got = m.SearchByProjection_Frame(cur, s.Tcw7, K_TUM3, s.last_keys, s.last_mp, s.last_outlier, s.mp_pos, s.mp_desc, s.mp_obs, cur_mp0, 15.0)
    n_ref, ref = O.search_by_projection_frame(s.cur_keys, s.cur_desc, s.w, s.h, s.sf, s.Tcw7, K_TUM3, s.last_keys, s.last_mp, s.last_outlier, s.mp_pos,
                                              s.mp_desc, s.mp_obs, cur_mp0, 15.0, True)
    assert n_gpu == n_ref and np.array_equal(got, ref)


@pytest.mark.parametrize("seed,th", [(0, 1.0), (1, 3.0), (2, 5.0), (4, 15.0)])
def test_search_by_projection_mappoints(matcher, seed, th):
    from rumi_slam_amd.matcher import FrameView
    s = TrackingScene(seed)
    F = FrameView(s.cur_keys, s.cur_desc, s.w, s.h, s.sf)
    mp = s.mappoint_view()
    # some features already hold a map point (as after TrackWithMotionModel): exercises the occupancy rule
    rng = np.random.default_rng(seed)
    frame_mp = np.where(rng.random(F.n) < 0.15, rng.integers(0, len(mp["obs"]), F.n), -1).astype(np.int32)
    n_ref, ref = O.search_by_projection_mappoints(s.cur_keys, s.cur_desc, s.w, s.h, s.sf, mp, frame_mp, th, True, 40.0, 0.8)
    n_gpu, got = matcher(0.8).SearchByProjection_MapPoints(F, mp, frame_mp, th, True, 40.0)
    assert n_ref > 50
    assert n_gpu == n_ref
    assert np.array_equal(got, ref), f"{np.count_nonzero(got != ref)} features differ"


@pytest.mark.parametrize("seed,nn", [(0, 0.7), (1, 0.75), (5, 0.9)])
def test_search_by_bow(matcher, seed, nn):
    from rumi_slam_amd.matcher import FrameView
    s = TrackingScene(seed)
    KF = FrameView(s.last_keys, s.last_desc, s.w, s.h, s.sf)
    F = FrameView(s.cur_keys, s.cur_desc, s.w, s.h, s.sf)
    fv_kf, fv_f = s.feature_vectors()
    a, b = _fv(fv_kf), _fv(fv_f)
    mp_bad = (np.random.default_rng(seed).random(len(s.mp_obs)) < 0.05).astype(np.uint8)
    for ori in (True, False):
        n_ref, ref = O.search_by_bow(s.last_keys, s.last_desc, s.last_mp, mp_bad, (a.node_ids, a.offsets, a.indices), s.cur_keys, s.cur_desc,
                                     (b.node_ids, b.offsets, b.indices), nn, ori)
        n_gpu, got = matcher(nn, ori).SearchByBoW(KF, a, s.last_mp, mp_bad, F, b)
        assert n_ref > 50
        assert n_gpu == n_ref
        assert np.array_equal(got, ref)


@pytest.mark.parametrize("K,nodes", [(10, 600), (4, 60), (1, 25), (3, 1)])
def test_search_by_bow_batch_matches_single_calls(matcher, K, nodes):
    """rumi_search_by_bow_batch: K candidate key-frames against one frame in one launch == K single SearchByBoW calls == the oracle, per
    candidate; `nodes` small puts dozens of features into every FeatureVector node (long sequential chains inside a node); nodes = 1 puts ALL
    of the frame's features into one node -- more than the 512 the batch kernel holds per node, so the entry answers with K single searches."""
    from rumi_slam_amd.matcher import FrameView, SearchByBoW_batch
    scenes = [TrackingScene(40 + k) for k in range(K)]
    base = scenes[0]
    F = FrameView(base.cur_keys, base.cur_desc, base.w, base.h, base.sf)
    _, fv_f0 = base.feature_vectors(nodes)
    b = _fv(fv_f0)
    KFs, fvs, mps, bads = [], [], [], []
    for k, s in enumerate(scenes):
        # candidate k: its own key-frame features; descriptors of some features replaced by the frame's so that true matches exist
        kdesc = s.last_desc.copy()
        n = min(len(kdesc), len(base.cur_desc))
        take = np.random.default_rng(k).random(n) < 0.5
        kdesc[:n][take] = base.cur_desc[:n][take]
        node_kf = {}
        fnode = {int(i): nd for nd, idxs in fv_f0.items() for i in idxs}
        rng = np.random.default_rng(100 + k)
        for i in range(len(kdesc)):
            nd = fnode[i] if i < n and take[i] else int(rng.integers(0, nodes)) * 7 + 3
            node_kf.setdefault(nd, []).append(i)
        KFs.append(FrameView(s.last_keys, kdesc, s.w, s.h, s.sf)); fvs.append(_fv(node_kf)); mps.append(s.last_mp)
        bads.append((np.random.default_rng(k).random(len(s.mp_obs)) < 0.05).astype(np.uint8))
    for ori in (True, False):
        m = matcher(0.75, ori)
        nm, got = SearchByBoW_batch(m, KFs, fvs, mps, bads, F, b)
        for k in range(K):
            n_ref, ref = O.search_by_bow(KFs[k].keys, KFs[k].desc, mps[k], bads[k], (fvs[k].node_ids, fvs[k].offsets, fvs[k].indices), base.cur_keys,
                                         base.cur_desc, (b.node_ids, b.offsets, b.indices), 0.75, ori)
            n_one, one = m.SearchByBoW(KFs[k], fvs[k], mps[k], bads[k], F, b)
            assert n_ref > 30, "scene gives matches"
            assert nm[k] == n_ref == n_one, f"candidate {k}: count"
            assert np.array_equal(got[k], ref) and np.array_equal(one, ref), f"candidate {k}: {np.count_nonzero(got[k] != ref)} features differ"


def test_dense_conflicts_force_many_fixpoint_rounds(matcher):
    """Many identical descriptors in one window: every query wants the same feature, so the sequential 'already taken'
    rule cascades — the worst case for the parallel fix-point resolver."""
    from rumi_slam_amd.matcher import FrameView
    rng = np.random.default_rng(3)
    n = 400
    keys = np.zeros(n, O.KP_DTYPE)
    keys["x"] = 300 + rng.uniform(-6, 6, n); keys["y"] = 200 + rng.uniform(-6, 6, n); keys["octave"] = 0
    keys["angle"] = rng.uniform(0, 360, n)
    base = rng.integers(0, 256, 32, dtype=np.uint8)
    desc = np.tile(base, (n, 1)); desc[:, 0] = rng.integers(0, 4, n)            # near-duplicates
    sf = (1.2 ** np.arange(8)).astype(np.float32)
    F = FrameView(keys, desc, 640, 480, sf)
    nmp = 300
    mp = dict(track_in_view=np.ones(nmp, np.uint8), proj_x=np.full(nmp, 300, np.float32), proj_y=np.full(nmp, 200, np.float32),
              scale_level=np.zeros(nmp, np.int32), view_cos=np.full(nmp, 0.9, np.float32), track_depth=np.ones(nmp, np.float32),
              is_bad=np.zeros(nmp, np.uint8), desc=np.tile(base, (nmp, 1)), obs=np.ones(nmp, np.int32))
    frame_mp = np.full(n, -1, np.int32)
    n_ref, ref = O.search_by_projection_mappoints(keys, desc, 640, 480, sf, mp, frame_mp, 3.0, False, 0.0, 1.1)
    n_gpu, got = matcher(1.1).SearchByProjection_MapPoints(F, mp, frame_mp, 3.0)
    assert n_ref >= 200
    assert n_gpu == n_ref and np.array_equal(got, ref)


def test_empty_inputs(matcher):
    from rumi_slam_amd.matcher import FrameView
    s = TrackingScene(0)
    cur = FrameView(s.cur_keys, s.cur_desc, s.w, s.h, s.sf)
    n, got = matcher().SearchByProjection_Frame(cur, s.Tcw7, K_TUM3, s.last_keys[:0], s.last_mp[:0], s.last_outlier[:0], s.mp_pos, s.mp_desc,
                                                s.mp_obs, np.full(cur.n, -1, np.int32), 15.0)
    assert n == 0 and (got == -1).all()
    empty = FrameView(s.cur_keys[:0], s.cur_desc[:0], s.w, s.h, s.sf)
    n, got = matcher().SearchByProjection_Frame(empty, s.Tcw7, K_TUM3, s.last_keys, s.last_mp, s.last_outlier, s.mp_pos, s.mp_desc, s.mp_obs,
                                                np.zeros(0, np.int32), 15.0)
    assert n == 0


def test_bruteforce_batch():
    import torch
    from rumi_slam_amd.matcher import bruteforce_batch
    rng = np.random.default_rng(0)
    B, cap = 3, 1100
    q = rng.integers(0, 256, (B, cap, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (B, cap, 32), dtype=np.uint8)
    t[1, 5] = t[1, 900]                                    # exact duplicate: first index must win
    q[1, 0] = t[1, 900]
    cq = np.array([[1000, 0], [1100, 0], [37, 0]], np.int32)
    ct = np.array([[1005, 0], [1100, 0], [300, 0]], np.int32)
    bi, bd, sd = bruteforce_batch(torch.from_numpy(q).cuda(), torch.from_numpy(cq).cuda(), torch.from_numpy(t).cuda(), torch.from_numpy(ct).cuda())
    torch.cuda.synchronize()
    for b in range(B):
        rbi, rbd, rsd = O.bruteforce_match(q[b, :cq[b, 0]], t[b, :ct[b, 0]])
        n = cq[b, 0]
        assert np.array_equal(bi[b, :n].cpu().numpy(), rbi) and np.array_equal(bd[b, :n].cpu().numpy(), rbd)
        assert np.array_equal(sd[b, :n].cpu().numpy(), rsd)
    assert bi[1, 0].item() == 5 and bd[1, 0].item() == 0 and sd[1, 0].item() == 0
    # no train descriptors, and a train set that holds only the query's bitwise complement (distance 256 never wins: strict <)
    q2 = rng.integers(0, 256, (2, 8, 32), dtype=np.uint8)
    t2 = np.zeros((2, 8, 32), np.uint8); t2[1, 0] = ~q2[1, 0]
    bi, bd, sd = bruteforce_batch(torch.from_numpy(q2).cuda(), torch.from_numpy(np.array([[3, 0], [1, 0]], np.int32)).cuda(), torch.from_numpy(t2).cuda(),
                                  torch.from_numpy(np.array([[0, 0], [1, 0]], np.int32)).cuda())
    torch.cuda.synchronize()
    assert bi[0, :3].tolist() == [-1, -1, -1] and bd[0, :3].tolist() == [256] * 3 and sd[0, :3].tolist() == [256] * 3
    rbi, rbd, rsd = O.bruteforce_match(q2[1, :1], t2[1, :1])
    assert [bi[1, 0].item(), bd[1, 0].item(), sd[1, 0].item()] == [int(rbi[0]), int(rbd[0]), int(rsd[0])] == [-1, 256, 256]



def test_bruteforce_ring_equals_pairwise_calls():
    """rumi_match_bruteforce_ring_device (frame i against frame i + 1 of ONE buffer, the last against the first, one launch) gives what the
    pairwise batch call gives on the shifted views, strided frames included."""
    import torch
    from rumi_slam_amd.matcher import bruteforce_batch, bruteforce_ring
    rng = np.random.default_rng(31)
    for B, cap, stride_pad in [(1, 300, 0), (2, 257, 0), (5, 1000, 64), (9, 700, 0)]:
        buf = torch.from_numpy(rng.integers(0, 256, (B, cap * 32 + stride_pad), dtype=np.uint8)).cuda()
        desc = buf[:, :cap * 32].view(B, cap, 32) if stride_pad == 0 else buf.as_strided((B, cap, 32), (cap * 32 + stride_pad, 32, 1))
        counts = torch.from_numpy(np.stack([rng.integers(1, cap + 1, B), np.zeros(B)], 1).astype(np.int32)).cuda()
        ri, rd, rs = [t.cpu().numpy() for t in bruteforce_ring(desc, counts)]
        for b in range(B):
            t = (b + 1) % B
            bi, bd, sd = [x.cpu().numpy()[0] for x in bruteforce_batch(desc[b:b + 1].contiguous(), counts[b:b + 1], desc[t:t + 1].contiguous(), counts[t:t + 1])]
            nq = int(counts[b, 0])
            assert np.array_equal(ri[b, :nq], bi[:nq]) and np.array_equal(rd[b, :nq], bd[:nq]) and np.array_equal(rs[b, :nq], sd[:nq]), (B, b)

def test_bruteforce_on_extractor_records():
    """The bench path: consecutive frames extracted into per-frame records on the device and matched in place through the strided entry
    (descriptors of real key-points: many near-duplicates and ties, unlike random bytes)."""
    import torch
    from rumi_slam_amd import rumination as R
    from rumi_slam_amd.extractor import ORBextractor
    from rumi_slam_amd.matcher import bruteforce_batch
    from rumi_slam_amd.synth import synth_frame, warp_frame
    img0 = synth_frame(77)
    frames = [img0] + [warp_frame(img0, 100 + i)[0] for i in range(3)]
    ext = ORBextractor(1000, 1.2, 8, 20, 7, max_batch=4)
    cap = ext.cap if hasattr(ext, "cap") else 1000 + 4 * 8 + 64
    rec = ext.extract_batch_records(torch.from_numpy(np.stack(frames)).cuda(), cap=cap)
    ext.sync()
    kp, desc, counts = R.record_views(rec, cap)
    bi, bd, sd = bruteforce_batch(desc[1:], counts[1:], desc[:-1], counts[:-1])
    torch.cuda.synchronize()
    d, c = desc.cpu().numpy(), counts.cpu().numpy()
    for b in range(3):
        nq, nt = int(c[b + 1, 0]), int(c[b, 0])
        rbi, rbd, rsd = O.bruteforce_match(np.ascontiguousarray(d[b + 1, :nq]), np.ascontiguousarray(d[b, :nt]))
        assert np.array_equal(bi[b, :nq].cpu().numpy(), rbi) and np.array_equal(bd[b, :nq].cpu().numpy(), rbd) and np.array_equal(sd[b, :nq].cpu().numpy(), rsd), b
        assert (rbd < 40).sum() > 300, "consecutive warped frames are supposed to share most features"


@pytest.mark.parametrize("seed,nn", [(0, 0.75), (2, 0.9)])
def test_search_by_bow_keyframe_keyframe(matcher, seed, nn):
    from rumi_slam_amd.matcher import FrameView, SearchByBoW_KF
    s = TrackingScene(seed)
    KF1 = FrameView(s.last_keys, s.last_desc, s.w, s.h, s.sf)
    KF2 = FrameView(s.cur_keys, s.cur_desc, s.w, s.h, s.sf)
    fv1, fv2 = s.feature_vectors()
    a, b = _fv(fv1), _fv(fv2)
    rng = np.random.default_rng(seed)
    nmp = len(s.mp_obs) + 500
    kf2_mp = np.where(rng.random(KF2.n) < 0.85, rng.integers(0, nmp, KF2.n), -1).astype(np.int32)
    mp_bad = (rng.random(nmp) < 0.05).astype(np.uint8)
    for ori in (True, False):
        n_ref, ref = O.search_by_bow_kf(s.last_keys, s.last_desc, s.last_mp, (a.node_ids, a.offsets, a.indices), s.cur_keys, s.cur_desc, kf2_mp,
                                        (b.node_ids, b.offsets, b.indices), mp_bad, nn, ori)
        n_gpu, got = SearchByBoW_KF(matcher(nn, ori), KF1, a, s.last_mp, KF2, b, kf2_mp, mp_bad)
        assert n_ref > 30
        assert n_gpu == n_ref and np.array_equal(got, ref)


@pytest.mark.parametrize("seed,th,variant", [(0, 8, 0), (1, 4, 1), (3, 10, 1)])
def test_search_by_projection_sim3(matcher, seed, th, variant):
    from rumi_slam_amd.matcher import FrameView, SearchByProjection_Sim3
    s = TrackingScene(seed)
    g = s.point_geometry()
    KF = FrameView(s.cur_keys, s.cur_desc, s.w, s.h, s.sf)
    rng = np.random.default_rng(seed)
    n = len(s.mp_pos)
    pts = dict(skip=(rng.random(n) < 0.1).astype(np.uint8), pos=s.mp_pos, normal=g["normal"], min_dist=g["min_dist"], max_dist=g["max_dist"],
               desc=s.mp_desc)
    matched0 = np.where(rng.random(KF.n) < 0.1, -2, -1).astype(np.int32)
    log_sf = float(np.log(np.float32(1.2)))
    n_ref, ref = O.search_by_projection_sim3(s.cur_keys, s.cur_desc, s.w, s.h, s.sf, log_sf, s.Tcw7, g["Ow"], K_TUM3, pts, matched0, th, 1.0, variant)
    n_gpu, got = SearchByProjection_Sim3(matcher(), KF, log_sf, s.Tcw7, g["Ow"], K_TUM3, pts, matched0, th, 1.0, bool(variant))
    assert n_ref > 50
    assert n_gpu == n_ref and np.array_equal(got, ref), f"{np.count_nonzero(got != ref)} differ"


@pytest.mark.parametrize("seed,th,dist", [(0, 10.0, 100), (2, 3.0, 64)])
def test_search_by_projection_relocalisation(matcher, seed, th, dist):
    from rumi_slam_amd.matcher import FrameView, SearchByProjection_Reloc
    s = TrackingScene(seed)
    g = s.point_geometry()
    Cur = FrameView(s.cur_keys, s.cur_desc, s.w, s.h, s.sf)
    rng = np.random.default_rng(seed)
    n = len(s.mp_pos)
    pts = dict(skip=(rng.random(n) < 0.1).astype(np.uint8), pos=s.mp_pos, min_dist=g["min_dist"], max_dist=g["max_dist"], desc=s.mp_desc)
    cur_mp0 = np.where(rng.random(Cur.n) < 0.2, rng.integers(0, n, Cur.n), -1).astype(np.int32)
    log_sf = float(np.log(np.float32(1.2)))
    for ori in (True, False):
        n_ref, ref = O.search_by_projection_reloc(s.cur_keys, s.cur_desc, s.w, s.h, s.sf, log_sf, s.Tcw7, g["Ow"], K_TUM3, s.last_keys, s.last_mp, pts,
                                                  cur_mp0, th, dist, ori)
        n_gpu, got = SearchByProjection_Reloc(matcher(0.9, ori), Cur, log_sf, s.Tcw7, g["Ow"], K_TUM3, s.last_keys, s.last_mp, pts, cur_mp0, th, dist)
        assert n_ref > 50
        assert n_gpu == n_ref and np.array_equal(got, ref), f"{np.count_nonzero(got != ref)} differ"


@pytest.mark.parametrize("seed", [0, 3])
def test_is_in_frustum_then_search_local_points(matcher, seed):
    """SearchLocalPoints = Frame::isInFrustum for every local point, then SearchByProjection(F, points): both steps on the GPU."""
    from rumi_slam_amd.matcher import FrameView, isInFrustum
    from scene import quat_rotate
    s = TrackingScene(seed)
    g = s.point_geometry()
    _, R = quat_rotate(s.Tcw7[:4].astype(np.float64), np.zeros((1, 3)))
    R32 = R.astype(np.float32)
    pts = dict(pos=s.mp_pos, normal=g["normal"], min_dist=g["min_dist"], max_dist=g["max_dist"])
    log_sf = float(np.log(np.float32(1.2)))
    ref = O.is_in_frustum(R32.ravel(), s.Tcw7[4:], g["Ow"], K_TUM3, s.w, s.h, log_sf, 8, 0.5, pts)
    m = matcher(0.8)
    got = isInFrustum(m, R32.ravel(), s.Tcw7[4:], g["Ow"], K_TUM3, s.w, s.h, log_sf, 8, 0.5, pts)
    assert ref["track_in_view"].sum() > 300
    for k in ref:
        assert np.array_equal(got[k], ref[k]), k
    F = FrameView(s.cur_keys, s.cur_desc, s.w, s.h, s.sf)
    mp = dict(got, is_bad=np.zeros(len(s.mp_obs), np.uint8), desc=s.mp_desc, obs=s.mp_obs)
    frame_mp = np.full(F.n, -1, np.int32)
    n_ref, fm_ref = O.search_by_projection_mappoints(s.cur_keys, s.cur_desc, s.w, s.h, s.sf, mp, frame_mp, 3.0, False, 0.0, 0.8)
    n_gpu, fm = m.SearchByProjection_MapPoints(F, mp, frame_mp, 3.0)
    assert n_gpu == n_ref and n_ref > 100 and np.array_equal(fm, fm_ref)


@pytest.mark.parametrize("seed,th", [(0, 3.0), (2, 1.0), (4, 15.0)])
def test_search_local_points_fused(matcher, seed, th):
    """rumi_search_local_points: frustum test and projection search in one call, the per-point fields never leaving the device in
    between.  Against the oracle's two steps: same fields (skipped points untouched = not in view), same nToMatch, same matches; some
    points are skipped (already matched in this frame / bad) and some frame features already hold a map point."""
    from rumi_slam_amd.matcher import FrameView
    from scene import quat_rotate
    s = TrackingScene(seed)
    g = s.point_geometry()
    _, R = quat_rotate(s.Tcw7[:4].astype(np.float64), np.zeros((1, 3)))
    R32 = R.astype(np.float32)
    rng = np.random.default_rng(seed)
    n = len(s.mp_obs)
    skip = (rng.random(n) < 0.15).astype(np.uint8)
    pts = dict(pos=s.mp_pos, normal=g["normal"], min_dist=g["min_dist"], max_dist=g["max_dist"], desc=s.mp_desc, obs=s.mp_obs, skip=skip)
    log_sf = float(np.log(np.float32(1.2)))
    ref = O.is_in_frustum(R32.ravel(), s.Tcw7[4:], g["Ow"], K_TUM3, s.w, s.h, log_sf, 8, 0.5, pts)
    for k in ref:                                           # the reference never calls isInFrustum on the skipped points
        ref[k] = np.where(skip != 0, np.array(-1 if k in ("proj_x", "proj_y") else 0, ref[k].dtype), ref[k])
    F = FrameView(s.cur_keys, s.cur_desc, s.w, s.h, s.sf)
    frame_mp = np.full(F.n, -1, np.int32)
    frame_mp[rng.choice(F.n, 40, replace=False)] = rng.integers(0, n, 40)      # features that already hold a (matched) map point
    mp = dict(ref, is_bad=skip, desc=s.mp_desc, obs=s.mp_obs)
    n_ref, fm_ref = O.search_by_projection_mappoints(s.cur_keys, s.cur_desc, s.w, s.h, s.sf, mp, frame_mp, th, False, 0.0, 0.8)
    m = matcher(0.8)
    nto, n_gpu, fm, got = m.SearchLocalPoints(F, R32.ravel(), s.Tcw7[4:], g["Ow"], K_TUM3, log_sf, 8, pts, frame_mp, th)
    for k in ref:
        assert np.array_equal(got[k], ref[k]), k
    assert nto == int(ref["track_in_view"].sum()) and nto > 200
    assert n_gpu == n_ref and n_ref > 50 and np.array_equal(fm, fm_ref)
    # nothing in view (all skipped): no search, the frame's vector untouched
    pts0 = dict(pts, skip=np.ones(n, np.uint8))
    nto, n_gpu, fm, got = m.SearchLocalPoints(F, R32.ravel(), s.Tcw7[4:], g["Ow"], K_TUM3, log_sf, 8, pts0, frame_mp, th)
    assert nto == 0 and n_gpu == 0 and np.array_equal(fm, frame_mp) and not got["track_in_view"].any()


@pytest.mark.parametrize("seed,ratio,ori,window", [(0, 0.9, True, 100), (1, 0.9, False, 100), (2, 0.7, True, 40), (3, 1.5, True, 200)])
def test_search_for_initialization(matcher, seed, ratio, ori, window):
    """Monocular initialisation matcher: sequential steal rule (vMatchedDistance) replayed exactly; second call reuses vbPrevMatched."""
    from rumi_slam_amd.matcher import FrameView
    s = TrackingScene(seed)
    m = matcher(ratio, ori)
    F1 = FrameView(s.last_keys, s.last_desc, s.w, s.h, s.sf)
    F2 = FrameView(s.cur_keys, s.cur_desc, s.w, s.h, s.sf)
    pm0 = np.stack([s.last_keys["x"], s.last_keys["y"]], 1).astype(np.float32)
    n_ref, m_ref, pm_ref = O.search_for_initialization(s.last_keys, s.last_desc, s.cur_keys, s.cur_desc, s.w, s.h, pm0, window, ratio, ori)
    n_gpu, m12, pm = m.SearchForInitialization(F1, F2, pm0, window)
    assert n_ref > 50
    assert n_gpu == n_ref and np.array_equal(m12, m_ref) and np.array_equal(pm, pm_ref)
    # Tracking::MonocularInitialization calls it again on the next frame with the updated vbPrevMatched
    n_ref2, m_ref2, pm_ref2 = O.search_for_initialization(s.last_keys, s.last_desc, s.cur_keys, s.cur_desc, s.w, s.h, pm_ref, window, ratio, ori)
    n_gpu2, m12b, pmb = m.SearchForInitialization(F1, F2, pm, window)
    assert n_gpu2 == n_ref2 and np.array_equal(m12b, m_ref2) and np.array_equal(pmb, pm_ref2)


@pytest.mark.parametrize("seed,coarse,ori", [(0, False, True), (1, False, False), (2, True, True), (3, False, True)])
def test_search_for_triangulation(matcher, seed, coarse, ori):
    """LocalMapping::CreateNewMapPoints matcher: BoW node walk + epipole / epipolar-line tests, pairs identical to the oracle."""
    from rumi_slam_amd.matcher import FrameView, SearchForTriangulation
    s = TrackingScene(seed)
    KF1 = FrameView(s.last_keys, s.last_desc, s.w, s.h, s.sf)
    KF2 = FrameView(s.cur_keys, s.cur_desc, s.w, s.h, s.sf)
    fv1, fv2 = s.feature_vectors(n_nodes=150)
    a, b = _fv(fv1), _fv(fv2)
    rng = np.random.default_rng(seed)
    mp1 = np.where(rng.random(KF1.n) < 0.4, 1, -1).astype(np.int32)        # only un-tracked key-points are triangulated
    mp2 = np.where(rng.random(KF2.n) < 0.4, 1, -1).astype(np.int32)
    F12, ep = s.epipolar_geometry()
    m = matcher(0.6, ori)
    n_ref, ref = O.search_for_triangulation(s.last_keys, s.last_desc, mp1, (a.node_ids, a.offsets, a.indices), s.cur_keys, s.cur_desc, mp2,
                                            (b.node_ids, b.offsets, b.indices), s.sf, F12, ep, False, coarse, ori)
    n_gpu, got = SearchForTriangulation(m, KF1, a, mp1, KF2, b, mp2, F12, ep, False, coarse)
    assert n_ref > 40 and len(ref) == n_ref
    assert n_gpu == n_ref and np.array_equal(got, ref)
    if not coarse:      # the epipolar test must actually reject something, or the case proves nothing
        n_c, _ = O.search_for_triangulation(s.last_keys, s.last_desc, mp1, (a.node_ids, a.offsets, a.indices), s.cur_keys, s.cur_desc, mp2,
                                            (b.node_ids, b.offsets, b.indices), s.sf, F12, ep, False, True, False)
        n_f, _ = O.search_for_triangulation(s.last_keys, s.last_desc, mp1, (a.node_ids, a.offsets, a.indices), s.cur_keys, s.cur_desc, mp2,
                                            (b.node_ids, b.offsets, b.indices), s.sf, F12, ep, False, False, False)
        assert n_f < n_c
    n0, p0 = SearchForTriangulation(m, KF1, a, mp1, KF2, b, mp2, F12, ep, True, coarse)      # bOnlyStereo on mono key-frames
    assert n0 == 0 and len(p0) == 0


@pytest.mark.parametrize("seed,th,reproj", [(0, 3.0, True), (1, 3.0, False), (2, 4.0, False), (3, 2.5, True)])
def test_fuse_candidates(matcher, seed, th, reproj):
    """Search half of both Fuse overloads (LocalMapping::SearchInNeighbors th=3; loop/merge fusion th=3..4)."""
    from rumi_slam_amd.matcher import FrameView, FuseCandidates
    s = TrackingScene(seed)
    g = s.point_geometry()
    KF = FrameView(s.cur_keys, s.cur_desc, s.w, s.h, s.sf)
    rng = np.random.default_rng(seed)
    n = len(s.mp_pos)
    pts = dict(skip=(rng.random(n) < 0.1).astype(np.uint8), pos=s.mp_pos, normal=g["normal"], min_dist=g["min_dist"], max_dist=g["max_dist"],
               desc=s.mp_desc)
    log_sf = float(np.log(np.float32(1.2)))
    ref = O.fuse_candidates(s.cur_keys, s.cur_desc, s.w, s.h, s.sf, log_sf, s.Tcw7, g["Ow"], K_TUM3, pts, th, reproj)
    got = FuseCandidates(matcher(), KF, log_sf, s.Tcw7, g["Ow"], K_TUM3, pts, th, reproj)
    assert np.count_nonzero(ref >= 0) > 50
    assert np.array_equal(got, ref), f"{np.count_nonzero(got != ref)} differ"
    if reproj:          # the chi2 gate must bite, or the flag is untested
        loose = O.fuse_candidates(s.cur_keys, s.cur_desc, s.w, s.h, s.sf, log_sf, s.Tcw7, g["Ow"], K_TUM3, pts, th, False)
        assert np.count_nonzero(loose != ref) > 0


@pytest.mark.parametrize("seed,th", [(0, 7.5), (1, 7.5), (2, 3.0)])
def test_search_by_sim3(matcher, seed, th):
    """Loop/merge Sim3 refinement matcher: two window searches from camera-frame points + mutual agreement."""
    from rumi_slam_amd.matcher import FrameView, SearchBySim3
    s = TrackingScene(seed)
    KF1 = FrameView(s.last_keys, s.last_desc, s.w, s.h, s.sf)
    KF2 = FrameView(s.cur_keys, s.cur_desc, s.w, s.h, s.sf)
    rng = np.random.default_rng(100 + seed)
    fx, fy, cx, cy = [float(v) for v in K_TUM3]
    sf = s.sf.astype(np.float64)

    def side(keys, desc, u, v, z):
        pc = np.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], 1)
        dist = np.linalg.norm(pc, axis=1)
        mx = dist * sf[keys["octave"]]
        return dict(skip=(rng.random(len(keys)) < 0.15).astype(np.uint8), pc=pc.astype(np.float32), min_dist=(mx / sf[-1]).astype(np.float32),
                    max_dist=mx.astype(np.float32), desc=desc)

    wx, wy, z = s.proj                                   # where the last-frame points fall in the current image
    side1 = side(s.last_keys, s.last_desc, wx, wy, z)
    H = np.eye(3); H[:s.A.shape[0], :] = s.A
    Hi = np.linalg.inv(H)
    x2 = np.stack([s.cur_keys["x"], s.cur_keys["y"], np.ones(len(s.cur_keys))], 0).astype(np.float64)
    x1 = Hi @ x2
    side2 = side(s.cur_keys, s.cur_desc, x1[0] / x1[2], x1[1] / x1[2], rng.uniform(1, 8, len(s.cur_keys)))
    log_sf = float(np.log(np.float32(1.2)))
    n_ref, ref = O.search_by_sim3(s.last_keys, s.last_desc, s.cur_keys, s.cur_desc, s.w, s.h, s.sf, log_sf, K_TUM3, side1, side2, th)
    n_gpu, got = SearchBySim3(matcher(), KF1, KF2, K_TUM3, log_sf, side1, side2, th)
    assert n_ref > 50 and n_ref == np.count_nonzero(ref >= 0)
    assert n_gpu == n_ref and np.array_equal(got, ref), f"{np.count_nonzero(got != ref)} differ"


def _random_frame(rng, n, w=640, h=480, clustered=False):
    """Random key-points (not from an image): uniform or piled into a few spots, random octaves / angles / descriptors."""
    k = np.zeros(n, O.KP_DTYPE)
    if clustered and n:
        c = rng.uniform([40, 40], [w - 40, h - 40], (6, 2))
        p = c[rng.integers(0, 6, n)] + rng.normal(0, 6, (n, 2))
        k["x"], k["y"] = np.clip(p[:, 0], 0, w - 1), np.clip(p[:, 1], 0, h - 1)
    else:
        k["x"], k["y"] = rng.uniform(0, w, n), rng.uniform(0, h, n)
    k["octave"] = rng.integers(0, 8, n)
    k["angle"] = rng.uniform(0, 360, n)
    k["size"] = 31; k["response"] = rng.integers(7, 255, n); k["class_id"] = -1
    return k, rng.integers(0, 256, (n, 32), dtype=np.uint8)


@pytest.mark.parametrize("case", ["empty_frame", "no_queries", "all_invalid", "max_features_clustered", "list_arena_grows"])
def test_matcher_edge_cases(case):
    """Empty and degenerate inputs, the largest frame the grid sort takes (8192 key-points, piled up so that windows hold
    hundreds of candidates and many queries fight for the same features), and the candidate-list arena growing mid-call."""
    from rumi_slam_amd.matcher import FrameView, ORBmatcher
    rng = np.random.default_rng(abs(hash(case)) % 1000)
    sf = np.float32(1.2) ** np.arange(8, dtype=np.float32)
    n_f = {"empty_frame": 0, "max_features_clustered": 8192}.get(case, 1500)
    keys, desc = _random_frame(rng, n_f, clustered=case in ("max_features_clustered", "list_arena_grows"))
    nq = 0 if case == "no_queries" else (3000 if case == "max_features_clustered" else 900)
    mp = dict(track_in_view=(np.zeros(nq, np.uint8) if case == "all_invalid" else (rng.random(nq) < 0.9).astype(np.uint8)),
              proj_x=rng.uniform(0, 640, nq).astype(np.float32), proj_y=rng.uniform(0, 480, nq).astype(np.float32),
              scale_level=rng.integers(0, 8, nq).astype(np.int32), view_cos=rng.uniform(0.99, 1.0, nq).astype(np.float32),
              track_depth=rng.uniform(1, 60, nq).astype(np.float32), is_bad=(rng.random(nq) < 0.05).astype(np.uint8),
              desc=rng.integers(0, 256, (nq, 32), dtype=np.uint8), obs=rng.integers(0, 5, nq).astype(np.int32))
    if n_f and nq:                                  # put queries where the key-points are and give them near-identical descriptors
        pick = rng.integers(0, n_f, nq)
        mp["proj_x"], mp["proj_y"] = (keys["x"][pick] + rng.normal(0, 2, nq)).astype(np.float32), (keys["y"][pick] + rng.normal(0, 2, nq)).astype(np.float32)
        mp["scale_level"] = np.clip(keys["octave"][pick] + rng.integers(0, 2, nq), 0, 7).astype(np.int32)
        mp["desc"] = desc[pick].copy()
        mp["desc"][:, 0] ^= rng.integers(0, 256, nq, dtype=np.uint8)
    if case == "list_arena_grows":                  # one pile, one octave: every query lists ~all 1500 key-points (1.3 M entries > 300 k arena)
        keys["x"], keys["y"] = 320 + rng.normal(0, 5, n_f), 240 + rng.normal(0, 5, n_f)
        keys["octave"] = 3
        mp["proj_x"], mp["proj_y"] = (320 + rng.normal(0, 3, nq)).astype(np.float32), (240 + rng.normal(0, 3, nq)).astype(np.float32)
        mp["scale_level"][:] = 3
    fm0 = np.where(rng.random(n_f) < 0.1, rng.integers(0, max(nq, 1), n_f), -1).astype(np.int32) if nq else np.full(n_f, -1, np.int32)
    th = 12.0 if case in ("max_features_clustered", "list_arena_grows") else 3.0
    small = case == "list_arena_grows"              # 64 + 65536/256 queries' worth of list entries: a dense call must outgrow it
    m = ORBmatcher(0.8, True, max_features=8192, max_queries=1024 if small else 16384)
    F = FrameView(keys, desc, 640, 480, sf)
    n_ref, fm_ref = O.search_by_projection_mappoints(keys, desc, 640, 480, sf, mp, fm0, th, False, 0.0, 0.8)
    for rep in range(2):                            # second call: the grown arena is reused
        n_gpu, fm = m.SearchByProjection_MapPoints(F, mp, fm0, th)
        assert n_gpu == n_ref and np.array_equal(fm, fm_ref), f"{case}: {np.count_nonzero(fm != fm_ref)} differ"
    if case in ("max_features_clustered", "list_arena_grows"):
        assert n_ref > 200
    if case in ("empty_frame", "no_queries", "all_invalid"):
        assert n_ref == 0
