"""GPU parity of the HIP ORB extractor against the CPU oracle, through the C ABI.
Bar: bit-exact key-point records (28 B each, order included), descriptors, counts and monoIndex."""
import numpy as np
import pytest

import oracle_lib
from rumi_slam_amd.synth import synth_frame

pytestmark = pytest.mark.gpu


def _pair(nf=1000, sf=1.2, nl=8, ini=20, mn=7, w=640, h=480, batch=1, blur_variant=0):
    from rumi_slam_amd.extractor import ORBextractor
    return (ORBextractor(nf, sf, nl, ini, mn, max_width=w, max_height=h, max_batch=batch, blur_variant=blur_variant),
            oracle_lib.OracleExtractor(nf, sf, nl, ini, mn, blur_variant=blur_variant))


def _assert_same(gpu_out, orc_out, tag=""):
    gm, gk, gd = gpu_out
    om, ok, od = orc_out
    assert len(gk) == len(ok), f"{tag}: key-point count {len(gk)} vs oracle {len(ok)}"
    assert gm == om, f"{tag}: monoIndex {gm} vs {om}"
    if gk.tobytes() != ok.tobytes():
        for name in gk.dtype.names:
            bad = np.nonzero(gk[name] != ok[name])[0]
            assert len(bad) == 0, f"{tag}: field {name} differs at {bad[:5]}: {gk[name][bad[:5]]} vs {ok[name][bad[:5]]}"
    assert np.array_equal(gd, od), f"{tag}: descriptors differ in {np.count_nonzero((gd != od).any(axis=1))} rows"


@pytest.mark.parametrize("seed", [1234, 1235, 1236, 7, 99])
def test_full_frame_bit_exact(seed):
    g, o = _pair()
    img = synth_frame(seed)
    _assert_same(g(img, None, (0, 1000)), o.extract(img, (0, 1000)), f"seed {seed}")


def test_stages_bit_exact():
    g, o = _pair()
    img = synth_frame(4321)
    g(img, None, (0, 1000))
    o.extract(img, (0, 1000))
    for l in range(8):
        assert np.array_equal(g.pyramid_level(l), o.level(l)), f"pyramid level {l}"
        assert np.array_equal(g.pyramid_level(l, blurred=True), o.level(l, blurred=True)), f"blurred level {l}"
        gc, oc = g.stage_keypoints(l, 0), o.keypoints(l, False)
        assert len(gc) == len(oc), f"candidates level {l}: {len(gc)} vs {len(oc)}"
        assert gc.tobytes() == oc.tobytes(), f"candidate list level {l}"
        gs, os_ = g.stage_keypoints(l, 1), o.keypoints(l, True)
        assert gs.tobytes() == os_.tobytes(), f"selected key-points level {l}"


def test_blur_variant_sepfilter_bit_exact():
    """RumiOrbConfig.blur_variant = 1 (the sepFilter2D GaussianBlur of OpenCV 3.4.0 / 3.4.1: taps {18,34,49,55,..}/256, saturated) in kernel and
    oracle alike: blurred levels, key-points and descriptors bit-exact; and the variant does change descriptors (it is a real switch)."""
    g1, o1 = _pair(blur_variant=1)
    g0, _ = _pair()
    img = synth_frame(4322)
    img[100:140, 200:260] = 255                                   # a saturated patch: 257 / 256 of 255 must clamp to 255
    out1 = g1(img, None, (0, 1000))
    _assert_same(out1, o1.extract(img, (0, 1000)), "blur variant 1")
    for l in range(8):
        assert np.array_equal(g1.pyramid_level(l, blurred=True), o1.level(l, blurred=True)), f"blurred level {l}, variant 1"
    out0 = g0(img, None, (0, 1000))
    assert out0[1].tobytes() == out1[1].tobytes(), "key-points do not depend on the blur"
    assert not np.array_equal(out0[2], out1[2]), "the two GaussianBlur variants give different descriptors"


def test_lapping_forward_order():
    # KFDSample passes {0,0}: every key-point takes the mono branch (forward order)   KFDSample.h:53
    g, o = _pair()
    img = synth_frame(55)
    _assert_same(g(img, None, (0, 0)), o.extract(img, (0, 0)), "lap {0,0}")
    _assert_same(g(img, None, (200, 400)), o.extract(img, (200, 400)), "lap {200,400}")


@pytest.mark.parametrize("nf", [2000, 5000])
def test_more_features(nf):
    g, o = _pair(nf=nf)
    img = synth_frame(1234 + nf)
    _assert_same(g(img, None, (0, 1000)), o.extract(img, (0, 1000)), f"nfeatures {nf}")


def test_low_texture_retry_path():
    # cells with no corner at iniThFAST fall back to minThFAST (ORBextractor.cc:783-785)
    g, o = _pair()
    for seed, kw in [(3, dict(n_rect=60, contrast=(8, 19))), (4, dict(n_rect=120, contrast=(8, 40))), (5, dict(n_rect=0))]:
        img = synth_frame(seed, **kw)
        out = o.extract(img, (0, 1000))
        _assert_same(g(img, None, (0, 1000)), out, f"low texture {seed}")


# the second row: widths whose levels put the row edge at a wave boundary of the blur (w mod 256 in 253..255 / 0..8), where it picks narrower strips
@pytest.mark.parametrize("wh", [(752, 480), (600, 350), (320, 240), (1241, 376),
                                (768, 480), (520, 400), (509, 381), (1025, 400), (515, 387)])
def test_other_sizes(wh):
    w, h = wh
    g, o = _pair(w=w, h=h)
    img = synth_frame(77, w=w, h=h)
    _assert_same(g(img, None, (0, 1000)), o.extract(img, (0, 1000)), f"size {wh}")


def test_strided_input_and_empty():
    g, o = _pair()
    big = np.zeros((480, 701), np.uint8)
    big[:, 3:643] = synth_frame(8)
    view = big[:, 3:643]                                      # row pitch 701 (odd), first pixel at an odd address
    assert not view.flags["C_CONTIGUOUS"]
    _assert_same(g(view, None, (0, 1000)), o.extract(np.ascontiguousarray(view), (0, 1000)), "strided")
    mono, k, d = g(np.zeros((0, 0), np.uint8))
    assert mono == -1 and len(k) == 0


@pytest.mark.parametrize("off,pitch", [(3, 701), (4, 704), (0, 641)])
def test_batch_device_strided_views(off, pitch):
    """rumi_orb_extract_batch_device on frames that are views into a wider buffer: unaligned base / odd pitch (staged through the aligned
    arena) and an aligned view (read in place); every frame must come out as it does from a dense copy."""
    import torch
    from rumi_slam_amd.synth import synth_batch
    B = 5
    g, o = _pair(batch=B)
    frames = synth_batch(B, seed0=640)
    wide = torch.zeros((B, 481, pitch), dtype=torch.uint8, device="cuda")
    wide[:, :480, off:off + 640] = torch.from_numpy(frames).cuda()
    view = wide[:, :480, off:off + 640]
    assert not view.is_contiguous() and view.stride(1) == pitch
    kp, desc, counts = g.extract_batch(view)
    torch.cuda.synchronize()
    kp = kp.cpu().numpy(); desc = desc.cpu().numpy(); counts = counts.cpu().numpy()
    for f in range(B):
        om, ok, od = o.extract(frames[f], (0, 1000))
        n = counts[f, 0]
        gk = kp[f, :n].copy().view(oracle_lib.KP_DTYPE).reshape(-1)
        _assert_same((int(counts[f, 1]), gk, desc[f, :n]), (om, ok, od), f"strided batch frame {f} off {off} pitch {pitch}")


def test_async_batches_match_blocking_calls():
    """Three batches enqueued back to back without waiting (rumi_orb_extract_batch_device_async, scratch arenas shared in stream order),
    one rumi_orb_sync at the end: every batch must equal what the blocking call gives."""
    import torch
    from rumi_slam_amd.synth import synth_batch
    B = 70
    g, _ = _pair(batch=B)
    batches = [torch.from_numpy(synth_batch(B, seed0=900 + 100 * i)).cuda() for i in range(3)]
    ref = []
    for fr in batches:
        kp, desc, counts = g.extract_batch(fr)
        ref.append((kp.cpu().numpy(), desc.cpu().numpy(), counts.cpu().numpy()))
    outs = [g.extract_batch(fr, wait=False) for fr in batches]
    g.sync()
    torch.cuda.synchronize()
    for i, (kp, desc, counts) in enumerate(outs):
        c = counts.cpu().numpy()
        assert np.array_equal(c, ref[i][2]), f"batch {i}: counts"
        k, d = kp.cpu().numpy(), desc.cpu().numpy()
        for f in range(B):
            n = c[f, 0]
            assert k[f, :n].tobytes() == ref[i][0][f, :n].tobytes(), f"batch {i} frame {f}: key-points"
            assert np.array_equal(d[f, :n], ref[i][1][f, :n]), f"batch {i} frame {f}: descriptors"

@pytest.mark.gpu
def test_resident_queue_with_unaligned_frames_back_to_back():
    """Advisor finding of round 3: with the resident queue on, frames whose width / pitch is not a multiple of 4 are first copied into an aligned
    staging arena by copies queued on the CALLER's stream -- which the slot streams of a resident call do not wait for from the second
    back-to-back call on.  Three different 641-pixel-wide batches enqueued without waiting must give what the blocking calls give (before the
    fix the later calls read the staging arena before their own copies had landed: the previous call's frames)."""
    import torch
    from rumi_slam_amd.synth import synth_batch
    W, H, n = 641, 480, 40
    g, _ = _pair(w=W, h=H, batch=n)
    batches = [torch.from_numpy(synth_batch(n, seed0=8100 + 100 * i, w=W, h=H)).cuda() for i in range(3)]
    ref = []
    for fr in batches:
        kp, desc, counts = g.extract_batch(fr)
        ref.append((kp.cpu().numpy(), desc.cpu().numpy(), counts.cpu().numpy()))
    assert not np.array_equal(ref[0][2], ref[1][2]), "the batches must differ"
    g.set_resident_queue(True)
    cap = 1000 + 4 * 8 + 64
    outs = [(torch.empty((n, cap, 7), dtype=torch.float32, device="cuda"), torch.empty((n, cap, 32), dtype=torch.uint8, device="cuda"),
             torch.empty((n, 2), dtype=torch.int32, device="cuda")) for _ in batches]
    torch.cuda.synchronize()
    for rep in range(2):
        for fr, o in zip(batches, outs):
            g.extract_batch(fr, wait=False, out=o)
        g.sync()
        torch.cuda.synchronize()
        for i, (kp, desc, counts) in enumerate(outs):
            c = counts.cpu().numpy()
            assert np.array_equal(c, ref[i][2]), f"round {rep} call {i}: counts"
            k, d = kp.cpu().numpy(), desc.cpu().numpy()
            for f in range(n):
                m = c[f, 0]
                assert k[f, :m].tobytes() == ref[i][0][f, :m].tobytes() and np.array_equal(d[f, :m], ref[i][1][f, :m]), f"round {rep} call {i} frame {f}"


def test_resident_queue_calls_overlap_and_match_blocking_calls():
    """rumi_orb_set_resident_queue: sub-chunks never wait for the caller's stream, consecutive calls rotate through four slots and run side by
    side.  Calls of different sizes (one sub-chunk, an odd size, a single frame, three sub-chunks: a call is cut at 256 frames) enqueued back to
    back, one sync at the end: every call must equal what the ordinary blocking call gives; the taps serve the last sub-chunk."""
    import torch
    from rumi_slam_amd.synth import synth_batch
    B = 520
    g, _ = _pair(batch=B)
    sizes = [64, 64, 128, 64, 200, 37, 1, 64, 520]
    batches = [torch.from_numpy(synth_batch(n, seed0=7000 + 50 * i)).cuda() for i, n in enumerate(sizes)]
    ref = []
    for fr in batches:
        kp, desc, counts = g.extract_batch(fr)
        ref.append((kp.cpu().numpy(), desc.cpu().numpy(), counts.cpu().numpy()))
    g.set_resident_queue(True)
    cap = 1000 + 4 * 8 + 64
    def bufs(n):
        return (torch.empty((n, cap, 7), dtype=torch.float32, device="cuda"), torch.empty((n, cap, 32), dtype=torch.uint8, device="cuda"),
                torch.empty((n, 2), dtype=torch.int32, device="cuda"))
    outs = [bufs(n) for n in sizes]                             # preallocated: a resident-queue call may not depend on the caller's stream
    torch.cuda.synchronize()
    with pytest.raises(ValueError):
        g.extract_batch(batches[0], wait=False)                 # the wrapper refuses to allocate outputs at call time
    for rep in range(2):                                        # second round: the slots are in use when the calls arrive, the buffers are reused
        if rep == 1:
            for o in outs:
                for t in o: t.zero_()
            done = torch.cuda.Event(); done.record()
            g.wait_event(done)                                  # the first call of the round starts behind the clearing of ALL buffers
        for fr, o in zip(batches, outs):
            g.extract_batch(fr, wait=False, out=o)
        g.sync()
        torch.cuda.synchronize()
        for i, (kp, desc, counts) in enumerate(outs):
            c = counts.cpu().numpy()
            assert np.array_equal(c, ref[i][2]), f"round {rep} call {i}: counts"
            k, d = kp.cpu().numpy(), desc.cpu().numpy()
            for f in range(sizes[i]):
                n = c[f, 0]
                assert k[f, :n].tobytes() == ref[i][0][f, :n].tobytes(), f"round {rep} call {i} frame {f}: key-points"
                assert np.array_equal(d[f, :n], ref[i][1][f, :n]), f"round {rep} call {i} frame {f}: descriptors"
    # taps: the last call had three sub-chunks of 174, 174 and 172 frames; frame 519 is in the last one, frame 0 is not
    lvl = g.pyramid_level(3, frame=519)
    g.set_resident_queue(False)
    g.extract_batch(batches[-1])
    assert np.array_equal(lvl, g.pyramid_level(3, frame=519))
    g.set_resident_queue(True)
    g.extract_batch(batches[-1], out=outs[-1])
    with pytest.raises(Exception):
        g.pyramid_level(3, frame=0)



@pytest.mark.parametrize("pinned", [False, True])
def test_host_batch_bit_exact_300_frames(pinned):
    """rumi_orb_extract_batch_host: 300 host frames (pageable: through the pinned staging slots; pinned: copied in place), transfers
    overlapped with the extraction group by group; every frame bit-exact against the oracle, device and host outputs alike."""
    import torch
    from rumi_slam_amd.synth import synth_batch
    B = 300
    g, o = _pair(batch=B)
    frames = synth_batch(B, seed0=3000)
    src = torch.from_numpy(frames).pin_memory() if pinned else [frames[f] for f in range(B)]
    # (pinned and pageable host outputs)
    (kp, desc, counts), (hk, hd, hc) = g.extract_batch_host(src, to_host="pinned" if pinned else True)
    torch.cuda.synchronize()
    kp = kp.cpu().numpy(); desc = desc.cpu().numpy(); counts = counts.cpu().numpy()
    assert np.array_equal(counts, hc)
    for f in range(B):
        om, ok, od = o.extract(frames[f], (0, 1000))
        n = counts[f, 0]
        gk = kp[f, :n].copy().view(oracle_lib.KP_DTYPE).reshape(-1)
        _assert_same((int(counts[f, 1]), gk, desc[f, :n]), (om, ok, od), f"host batch frame {f}")
        assert hk[f, :n].tobytes() == gk.tobytes() and np.array_equal(hd[f, :n], desc[f, :n]), f"host copies of frame {f}"


def test_batch_device_matches_single():
    import torch
    from rumi_slam_amd.synth import synth_batch
    B = 6
    g, o = _pair(batch=B)
    frames = synth_batch(B, seed0=500)
    kp, desc, counts = g.extract_batch(torch.from_numpy(frames).cuda())
    torch.cuda.synchronize()
    kp = kp.cpu().numpy(); desc = desc.cpu().numpy(); counts = counts.cpu().numpy()
    for f in range(B):
        om, ok, od = o.extract(frames[f], (0, 1000))
        n = counts[f, 0]
        gk = kp[f, :n].copy().view(oracle_lib.KP_DTYPE).reshape(-1)
        _assert_same((int(counts[f, 1]), gk, desc[f, :n]), (om, ok, od), f"batch frame {f}")


@pytest.mark.parametrize("B,max_batch", [(70, 70), (130, 256), (300, 300)])
def test_pipelined_batches_match_single_frames(B, max_batch):
    """Batches of 64 frames and more run as sub-chunks pipelined over four streams with scratch slots reused in stream order (70 = 2 x 35,
    130 = 4 x 33 and change, 300 = 5 sub-chunks of 64 over 4 slots): every frame must come out exactly as it does alone, and as the oracle
    has it."""
    import torch
    from rumi_slam_amd.synth import synth_batch
    g, o = _pair(batch=max_batch)
    g1, _ = _pair(batch=1)
    frames = synth_batch(B, seed0=800)
    for rep in range(2):                                    # twice: the second call reuses every scratch slot
        kp, desc, counts = g.extract_batch(torch.from_numpy(frames).cuda())
        torch.cuda.synchronize()
    kp = kp.cpu().numpy(); desc = desc.cpu().numpy(); counts = counts.cpu().numpy()
    for f in range(B):
        mono, k1, d1 = g1(frames[f])
        n = counts[f, 0]
        gk = kp[f, :n].copy().view(oracle_lib.KP_DTYPE).reshape(-1)
        _assert_same((int(counts[f, 1]), gk, desc[f, :n]), (mono, k1, d1), f"batch of {B}, frame {f}")
    for f in (0, B // 2, B - 1):
        om, ok, od = o.extract(frames[f], (0, 1000))
        n = counts[f, 0]
        _assert_same((int(counts[f, 1]), kp[f, :n].copy().view(oracle_lib.KP_DTYPE).reshape(-1), desc[f, :n]), (om, ok, od), f"batch of {B}, frame {f} vs oracle")


def test_golden_fixtures_on_gpu():
    import glob, os
    from rumi_slam_amd.extractor import ORBextractor
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for path in sorted(glob.glob(os.path.join(gold, "orb_*.npz"))):
        g = np.load(path)
        w, h = (int(v) for v in g["wh"])
        kw = dict(n_rect=60, contrast=(8, 19)) if "lowtex" in path else {}
        img = synth_frame(int(g["seed"]), w=w, h=h, **kw)
        ext = ORBextractor(int(g["nfeatures"]), 1.2, 8, 20, 7, max_width=w, max_height=h)
        mono, kps, desc = ext(img, None, tuple(int(v) for v in g["lap"]))
        assert mono == int(g["mono"]), path
        assert np.array_equal(kps.view(np.uint8).reshape(-1, 28), g["kps"]), path
        assert np.array_equal(desc, g["desc"]), path


def test_workgroup_sort_replays_std_sort():
    """The quadtree's fine rounds sort (size, UL.x) keys with std::sort; equal keys are frequent and their order is whatever
    libstdc++'s introsort leaves.  The workgroup-parallel replay (parallel Hoare partitions + windowed stable rank) must give
    the very same permutation as the real std::sort, including the heap-sort fallback on adversarial inputs."""
    from rumi_slam_amd import capi
    H = capi.hooks()
    rng = np.random.default_rng(7)
    cases = []
    for n in [0, 1, 2, 15, 16, 17, 18, 31, 33, 64, 65, 100, 257, 700, 1000, 2189, 4096]:
        for hi in (2, 5, 40, 100000):
            cases.append(rng.integers(0, hi, n).astype(np.uint32))
    for n in range(17, 65):                                                     # the one-wave form (entries in registers)
        for hi in (3, 12, 1000):
            cases.append(rng.integers(0, hi, n).astype(np.uint32))
        cases.append(np.arange(n, dtype=np.uint32))
        cases.append(np.arange(n, dtype=np.uint32)[::-1].copy())
        cases.append(np.concatenate([np.arange(n // 2), np.arange(n - n // 2)[::-1]]).astype(np.uint32))
        kk = np.zeros(n, np.uint32); hh = n // 2                                # median-of-3 killer at this size
        for i in range(hh):
            kk[2 * i] = i + 1 if i % 2 == 0 else 0
            kk[2 * i + 1] = hh + i + 1
        cases.append(kk)
    cases.append(np.arange(3000, dtype=np.uint32))                              # sorted
    cases.append(np.arange(3000, dtype=np.uint32)[::-1].copy())                # reversed
    cases.append(np.concatenate([np.arange(1500), np.arange(1500)[::-1]]).astype(np.uint32))   # organ pipe
    k = np.zeros(4096, np.uint32); k[::2] = np.arange(2048); k[1::2] = np.arange(2048)[::-1]
    cases.append(k)
    # median-of-3 killer (drives introsort into its depth limit -> heap-sort fallback)
    n = 2048; killer = np.zeros(n, np.uint32); h = n // 2
    for i in range(h):
        killer[2 * i] = i + 1 if i % 2 == 0 else 0
        killer[2 * i + 1] = h + i + 1
    cases.append(killer)
    for keys in cases:
        n = len(keys)
        ids = np.arange(n, dtype=np.uint16)
        k1, i1 = keys.copy(), ids.copy()
        k2, i2 = keys.copy(), ids.copy()
        assert H.rumi_hook_std_sort(capi.ptr(k1), capi.ptr(i1), n) == 0
        assert H.rumi_hook_sort_device(capi.ptr(k2), capi.ptr(i2), n) == 0
        assert np.array_equal(k1, k2), f"keys not sorted alike, n={n}"
        assert np.array_equal(i1, i2), f"tie order differs from std::sort, n={n}: {np.count_nonzero(i1 != i2)} positions"


def test_featureless_and_saturated_frames():
    """No corner anywhere (constant image), corners only from clipping (saturated blobs), and a frame with fewer candidates than
    wanted features on some levels: counts, order and descriptors still equal the oracle's; a constant image yields zero key-points."""
    g, o = _pair()
    flat = np.full((480, 640), 97, np.uint8)
    mono, k, d = g(flat, None, (0, 1000))
    om, ok, od = o.extract(flat, (0, 1000))
    assert len(k) == 0 and len(ok) == 0 and mono == om
    rng = np.random.default_rng(3)
    sat = np.zeros((480, 640), np.uint8)
    for _ in range(40):
        x, y, r = int(rng.integers(30, 610)), int(rng.integers(30, 450)), int(rng.integers(3, 12))
        sat[y - r:y + r, x - r:x + r] = 255
    _assert_same(g(sat, None, (0, 1000)), o.extract(sat, (0, 1000)), "saturated blobs")
    few = synth_frame(21, n_rect=6, noise=0)
    _assert_same(g(few, None, (0, 1000)), o.extract(few, (0, 1000)), "few candidates")


def test_level_with_more_keys_than_the_register_path_holds():
    """One 800 x 600 frame of white noise: level 0 alone has more than 24 x 512 FAST keys, so the quadtree workgroup of a single-frame
    call (register-resident keys, orb_octree_kernel.hip) takes its keys-in-memory route for that level and the register route
    for the upper ones; both must reproduce the oracle's selection and order."""
    g, o = _pair(nf=2000, w=800, h=600)
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (600, 800), dtype=np.uint8)
    gout, oout = g(img, None, (0, 1000)), o.extract(img, (0, 1000))
    assert len(g.stage_keypoints(0, 0)) > 24 * 512
    assert 0 < len(g.stage_keypoints(3, 0)) <= 24 * 512
    _assert_same(gout, oout, "noise 800x600")
    for l in range(8):
        assert g.stage_keypoints(l, 1).tobytes() == o.keypoints(l, True).tobytes(), f"selected key-points level {l}"


def _clustered_frame(seed, boxes):
    """Flat background, white noise inside a few small boxes (every FAST key of a level sits in a handful of quadtree cells) or, for the box
    "diag", inside a thin diagonal band: the list then grows by a factor of two per round, not four, and the tree is eight levels deep before it
    has the nodes wanted."""
    rng = np.random.default_rng(seed)
    img = np.full((480, 640), 120, np.uint8)
    for b in boxes:
        if b == "diag":
            yy, xx = np.mgrid[0:480, 0:640]
            band = np.abs(yy - 0.73 * xx - 5) < 9
            img[band] = rng.integers(0, 256, int(band.sum()), dtype=np.uint8)
            continue
        x0, y0, w, h = b
        img[y0:y0 + h, x0:x0 + w] = rng.integers(0, 256, (h, w), dtype=np.uint8)
    return img


@pytest.mark.parametrize("nf", [1000, 2000])
def test_clustered_corners_take_the_quadtree_past_its_count_tables(nf):
    """The quadtree kernel takes the quadrant counts of the first 4-6 subdivision levels from one histogram pass (count tables) and falls back
    to its relabel passes once a cell of the deepest table level is opened.  Corners clustered in a few small boxes force exactly that (the wanted
    number of nodes can only come from cells a few pixels wide): one frame (keys in registers), and batches of 40 frames (keys in memory: the
    256-thread kernel at 1000 features, the 512-thread one at 2000), against the oracle."""
    import torch
    g, o = _pair(nf=nf, batch=40)
    layouts = [["diag"], [(300, 200, 90, 70)], [(20, 20, 60, 60), (560, 400, 60, 60)], [(16, 16, 40, 440)], ["diag", (400, 60, 50, 45)],
               [(100, 100, 30, 30), (130, 130, 30, 30), (400, 90, 50, 45)], [(250, 180, 140, 120), (30, 400, 25, 25)]]
    frames = [_clustered_frame(40 + i, layouts[i % len(layouts)]) for i in range(40)]
    outs = [o.extract(f, (0, 1000)) for f in frames]
    assert max(len(k) for _, k, _ in outs) > 200, "the clusters must yield key-points"
    for i in range(len(layouts)):
        _assert_same(g(frames[i], None, (0, 1000)), outs[i], f"clustered layout {i}, one frame")
    cap = nf + 96
    kp, desc, counts = g.extract_batch(torch.from_numpy(np.stack(frames)).cuda(), (0, 1000), cap=cap)
    kp, desc, counts = kp.cpu().numpy(), desc.cpu().numpy(), counts.cpu().numpy()
    for i, (om, ok, od) in enumerate(outs):
        n = int(counts[i, 0])
        gk = kp[i, :n].copy().view(oracle_lib.KP_DTYPE).reshape(-1)
        _assert_same((int(counts[i, 1]), gk, desc[i, :n]), (om, ok, od), f"clustered batch frame {i}")


@pytest.mark.parametrize("ini,mn", [(20, 20), (7, 20), (30, 5), (12, 7), (40, 1)])
def test_threshold_pairs(ini, mn):
    """The two cv::FAST calls per cell (ORBextractor.cc:771-785) with unusual threshold pairs: equal, inverted (the retry can only find a subset
    of nothing), far apart, and minThFAST = 1; textured and low-texture frames (the latter live on the retry)."""
    g, o = _pair(ini=ini, mn=mn)
    for seed, kw in [(31, {}), (32, dict(n_rect=60, contrast=(8, 25), noise=1)), (33, dict(n_rect=12, contrast=(5, 14), noise=0))]:
        img = synth_frame(seed, **kw)
        _assert_same(g(img, None, (0, 1000)), o.extract(img, (0, 1000)), f"thresholds {ini}/{mn} seed {seed}")


@pytest.mark.gpu
def test_one_launch_pyramid_equals_the_launch_per_level_chain(tmp_path):
    """Calls of up to 4 frames compute the whole pyramid in one launch (k_pyramid_tiles: tiles of the top level, their regions of every level in
    LDS); the chain of one launch per level must produce the same bytes: 9 geometries (sizes, scale factors 1.1 / 1.2 / 1.5, 2 to 8 levels), calls
    of 1 and 3 frames, every level of every frame (tools/pyramid_tiles_check.py, once per path: the switch is read once per process)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dumps = []
    for tiles in ("0", "1"):
        out = str(tmp_path / f"tiles{tiles}.npz")
        subprocess.run([sys.executable, os.path.join(root, "tools", "pyramid_tiles_check.py"), out], check=True, env=dict(os.environ, RUMI_PYRAMID_TILES=tiles), timeout=600)
        dumps.append(np.load(out))
    a, b = dumps
    assert set(a.files) == set(b.files) and len(a.files) > 150
    bad = [k for k in a.files if not np.array_equal(a[k], b[k])]
    assert not bad, f"levels differ: {bad[:8]}"


@pytest.mark.gpu
def test_frame_captured_into_the_extractors_pinned_buffer():
    """rumi_orb_image_buffer: a frame written into the handle's pinned staging memory and passed by that pointer skips the staging copy; same result."""
    from rumi_slam_amd.extractor import ORBextractor
    from rumi_slam_amd.synth import synth_frame
    ext = ORBextractor(1000, 1.2, 8, 20, 7)
    img = synth_frame(4711)
    a = ext(img)
    buf = ext.image_buffer(640, 480)
    assert buf.shape == (480, 640) and buf.strides == (640, 1)
    buf[:] = img
    b = ext(buf)
    assert a[0] == b[0] and a[1].tobytes() == b[1].tobytes() and np.array_equal(a[2], b[2]) and len(a[1]) > 500
    assert ext.image_buffer(322, 240).strides == (324, 1)           # rows padded to 4 bytes
