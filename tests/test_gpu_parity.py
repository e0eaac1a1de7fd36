"""GPU parity: HIP path (through the C ABI) vs the CPU oracle, bit-exact on the same seeded inputs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CRIT_DEFAULT = (3, 30, 0.01)
CRIT_REF = (3, 25, 0.03)


def _pair(synth, w, h, ux, uy, seed=1234):
    return synth.frame(w, h, 0, 0, seed), synth.frame(w, h, ux, uy, seed)


@pytest.mark.parametrize("w,h", [(640, 480), (641, 479), (37, 19), (4, 3), (1023, 5)])
@pytest.mark.parametrize("variant", [3, 4])
def test_gray(ctx, orc, w, h, variant):
    rng = np.random.RandomState(w * 7 + h + variant)
    img = rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
    ctx.upload_bgr(0, img, variant)
    got = ctx.download_level(0, 0)
    assert np.array_equal(got, orc.bgr2gray(img, variant))


@pytest.mark.parametrize("w,h", [(640, 480), (641, 479), (333, 257), (64, 48), (23, 31), (100, 8), (1023, 767), (129, 257),
                                 (128, 128), (127, 385), (1000, 3), (1, 70)])
@pytest.mark.parametrize("ahead", [False, True])
def test_pyramid(ctx, orc, synth, w, h, ahead):
    """All levels, bit for bit: the one-launch kernel (three levels per launch, tiles with recomputed halos, reflection
    at every level's own border) -- sizes around the tile, odd sizes, levels narrower than a halo.  `ahead`: built on
    the copy stream (icelk_build_pyramid_ahead), where the one-wave geometry (64-px tiles) is used."""
    rng = np.random.RandomState(w + 13 * h)
    img = rng.randint(0, 256, size=(h, w)).astype(np.uint8)
    ctx.upload_gray(0, img)
    if ahead:
        ctx.build_pyramid_ahead(0, (5, 5), 6)
    top = ctx.build_pyramid(0, (5, 5), 6)
    ref = orc.build_pyramid(img, (5, 5), 6)
    assert top == len(ref) - 1
    for l, r in enumerate(ref):
        got = ctx.download_level(0, l)
        assert got.shape == r.shape
        assert np.array_equal(got, r), "level %d differs" % l


def test_pyramid_kernels_agree(synth, monkeypatch):
    """The per-level kernel (ICELK_PYR_PER_LEVEL=1) and the fused one are two statements of the same arithmetic."""
    from iceberg_tracking_code_amd import Context
    img = synth.frame(1000, 700, 77, -31, 5)
    out = []
    for per_level in (False, True):
        if per_level:
            monkeypatch.setenv("ICELK_PYR_PER_LEVEL", "1")
        c = Context(1000, 700, n_slots=1, max_pts=64)
        c.upload_gray(0, img)
        top = c.build_pyramid(0, (9, 9), 5)
        out.append([c.download_level(0, l) for l in range(top + 1)])
        c.close()
    assert len(out[0]) == len(out[1]) == 6
    for a, b in zip(*out):
        assert np.array_equal(a, b)


def test_pyramid_stop_rule(ctx, orc, synth):
    img = synth.frame(640, 480)
    ctx.upload_gray(0, img)
    for win, ml in [((21, 21), 3), ((35, 35), 4), ((35, 35), 10), ((100, 100), 5), ((400, 400), 3)]:
        assert ctx.build_pyramid(0, win, ml) == orc.pyramid_levels(640, 480, win, ml)


@pytest.mark.parametrize("w,h", [(640, 480), (200, 100), (37, 29)])
def test_synth_device_equals_numpy(ctx, synth, w, h):
    for ux, uy, seed in [(0, 0, 1234), (300, -200, 1234), (-7777, 12345, 99)]:
        ctx.synth_frame(1, w, h, ux, uy, seed)
        assert np.array_equal(ctx.download_level(1, 0), synth.frame(w, h, ux, uy, seed))
        for aff in ((0, 0, 0, 0), (5243, -3000, 2000, -5243), (-8192, 8192, 8192, -8192)):
            ctx.synth_frame(1, w, h, ux, uy, seed, affine=aff)
            assert np.array_equal(ctx.download_level(1, 0), synth.frame(w, h, ux, uy, seed, affine=aff)), aff
    with pytest.raises(ValueError):
        ctx.synth_frame(1, w, h, 0, 0, 1, affine=(9000, 0, 0, 0))


def _points(rng, n, w, h, border=-5.0):
    return np.stack([rng.uniform(border, w - border, n), rng.uniform(border, h - border, n)], 1).astype(np.float32)


@pytest.mark.parametrize("win,maxlevel,crit", [((21, 21), 3, CRIT_DEFAULT), ((35, 35), 4, CRIT_REF),
                                                ((31, 31), 5, CRIT_DEFAULT), ((5, 7), 2, CRIT_DEFAULT),
                                                ((15, 9), 0, (1, 5, 0.0)), ((11, 11), 3, (2, 0, 0.05)),
                                                ((41, 41), 2, CRIT_REF), ((64, 64), 1, CRIT_REF)])
def test_pyrlk_bit_exact(ctx, orc, synth, win, maxlevel, crit):
    w, h = 640, 480
    img0, img1 = _pair(synth, w, h, 410, -333)
    rng = np.random.RandomState(win[0] * 100 + maxlevel)
    # includes points near and beyond the borders (negative / > size coordinates)
    pts = _points(rng, 700, w, h, border=-12.0)
    ctx.upload_gray(0, img0)
    ctx.upload_gray(1, img1)
    p1, st, er = ctx.pyrlk(0, 1, pts, None, win, maxlevel, crit)
    q1, qs, qe = orc.pyrlk(img0, img1, pts, None, win, maxlevel, crit)
    assert np.array_equal(st, qs)
    assert np.array_equal(p1.view(np.uint32), q1.view(np.uint32)), "nextPts differ: max |d| = %g" % np.abs(p1 - q1).max()
    assert np.array_equal(er.view(np.uint32), qe.view(np.uint32))
    assert st.sum() > 0.8 * len(pts)


def test_pyrlk_flags(ctx, orc, synth):
    w, h = 320, 240
    img0, img1 = _pair(synth, w, h, -500, 250)
    rng = np.random.RandomState(5)
    pts = _points(rng, 300, w, h, 10.0)
    guess = pts + np.float32([1.5, -0.5])
    ctx.upload_gray(0, img0)
    ctx.upload_gray(1, img1)
    for flags in (4, 8, 12):
        p1, st, er = ctx.pyrlk(0, 1, pts, guess, (21, 21), 3, CRIT_DEFAULT, flags)
        q1, qs, qe = orc.pyrlk(img0, img1, pts, guess, (21, 21), 3, CRIT_DEFAULT, flags)
        assert np.array_equal(st, qs)
        assert np.array_equal(p1.view(np.uint32), q1.view(np.uint32))
        assert np.array_equal(er.view(np.uint32), qe.view(np.uint32))


def test_pyrlk_flat_and_outside(ctx, orc):
    """status must go to 0 on texture-less patches (minEig test) and for points far outside."""
    w, h = 200, 160
    img = np.full((h, w), 77, np.uint8)
    img[40:80, 50:120] = 200
    pts = np.float32([[10, 10], [150, 140], [50, 40], [119, 79], [-100, -100], [500, 500], [85, 60]])
    ctx.upload_gray(0, img)
    ctx.upload_gray(1, img)
    p1, st, er = ctx.pyrlk(0, 1, pts, None, (21, 21), 2, CRIT_DEFAULT)
    q1, qs, qe = orc.pyrlk(img, img, pts, None, (21, 21), 2, CRIT_DEFAULT)
    assert np.array_equal(st, qs) and np.array_equal(p1.view(np.uint32), q1.view(np.uint32))
    assert np.array_equal(er.view(np.uint32), qe.view(np.uint32))
    assert st.ravel()[4] == 0 and st.ravel()[5] == 0 and st.ravel()[0] == 0


def test_pyrlk_empty(ctx, synth):
    img = synth.frame(64, 48)
    ctx.upload_gray(0, img)
    ctx.upload_gray(1, img)
    p1, st, er = ctx.pyrlk(0, 1, np.zeros((0, 1, 2), np.float32))
    assert p1.shape == (0, 1, 2) and st.shape == (0, 1) and er.shape == (0, 1)


@pytest.mark.parametrize("win,maxlevel,crit", [((21, 21), 3, CRIT_DEFAULT), ((35, 35), 4, CRIT_REF)])
def test_track_fb_bit_exact(ctx, orc, synth, win, maxlevel, crit):
    w, h = 640, 480
    img0, img1 = _pair(synth, w, h, 600, 420)
    pts = orc.good_features(img0, 1500, 0.01, 7, blockSize=5).reshape(-1, 2)
    ctx.upload_gray(0, img0)
    ctx.upload_gray(1, img1)
    g = ctx.track_fb(0, 1, pts, win, maxlevel, crit)
    r = orc.track_fb(img0, img1, pts, win, maxlevel, crit)
    for k in ("p1", "p0r", "err_fwd", "err_bwd", "dist"):
        assert np.array_equal(g[k].view(np.uint32), r[k].view(np.uint32)), k
    for k in ("st_fwd", "st_bwd", "valid"):
        assert np.array_equal(g[k], r[k]), k
    flow = synth.true_flow((0, 0), (600, 420))
    ok = g["valid"].astype(bool)
    assert ok.mean() > 0.9
    e = np.abs((g["p1"] - pts)[ok] - flow).max(axis=1)
    assert np.median(e) < 0.05 and np.percentile(e, 95) < 0.15


def test_pyrlk_of_thirty_random_configurations(orc, synth):
    """The tracker kernels are chosen by window (15 / 21 / 31 / 35 square: the tuned one-feature-per-wave kernels; anything else:
    the generic one), and inside them tiles, segment tables and level loops depend on window, level count and frame size.
    Thirty random draws of all of that -- frames from 24 x 24 to 700 x 500 (smaller than a window, odd, not a multiple of
    anything), windows 3 .. 45 in each direction (even, odd, not square) with the tuned sizes over-represented, maxLevel 0 .. 5,
    every criteria type with counts 1 .. 40 and epsilons from 0 to 0.1, initial guesses, the minEig error flag, points inside,
    on and beyond the borders -- through icelk_pyrlk and through the fused forward + backward launch, every output bit for
    bit against the oracle."""
    from iceberg_tracking_code_amd import Context
    rng = np.random.RandomState(20261005)
    c = Context(704, 512, n_slots=2, max_pts=4096)
    tuned = [(15, 15), (21, 21), (31, 31), (35, 35)]
    try:
        for case in range(30):
            w, h = int(rng.randint(24, 701)), int(rng.randint(24, 501))
            win = tuned[rng.randint(4)] if rng.rand() < 0.4 else (int(rng.randint(3, 46)), int(rng.randint(3, 46)))
            maxlevel = int(rng.randint(0, 6))
            ctype = int(rng.randint(1, 4))
            crit = (ctype, int(rng.randint(1, 41)), float(rng.choice([0.0, 0.001, 0.01, 0.03, 0.1])))
            ux, uy = int(rng.randint(-600, 601)), int(rng.randint(-600, 601))
            img0, img1 = _pair(synth, w, h, ux, uy, seed=int(rng.randint(1, 10000)))
            pts = _points(rng, 300, w, h, border=-15.0)
            flags = int(rng.choice([0, 0, 4, 8, 12]))
            guess = (pts + rng.uniform(-2, 2, pts.shape).astype(np.float32)) if flags & 4 else None
            c.upload_gray(0, img0)
            c.upload_gray(1, img1)
            tag = (case, w, h, win, maxlevel, crit, flags)
            p1, st, er = c.pyrlk(0, 1, pts, guess, win, maxlevel, crit, flags)
            q1, qs, qe = orc.pyrlk(img0, img1, pts, guess, win, maxlevel, crit, flags)
            assert np.array_equal(st, qs), tag
            assert np.array_equal(p1.view(np.uint32), q1.view(np.uint32)), tag
            assert np.array_equal(er.view(np.uint32), qe.view(np.uint32)), tag
            g = c.track_fb(0, 1, pts, win, maxlevel, crit)
            r = orc.track_fb(img0, img1, pts, win, maxlevel, crit)
            for k in ("p1", "p0r", "err_fwd", "err_bwd", "dist"):
                assert np.array_equal(g[k].view(np.uint32), r[k].view(np.uint32)), (k,) + tag
            for k in ("st_fwd", "st_bwd", "valid"):
                assert np.array_equal(g[k], r[k]), (k,) + tag
    finally:
        c.close()


@pytest.mark.parametrize("bs", [3, 10, 5, 2])
@pytest.mark.parametrize("w,h", [(640, 480), (131, 77)])
def test_min_eig_map_bit_exact(ctx, orc, synth, w, h, bs):
    img = synth.frame(w, h, 17, 4242, 7)
    ctx.upload_gray(0, img)
    got = ctx.min_eig_map(0, bs)
    ref = orc.min_eig_map(img, bs)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), np.abs(got - ref).max()


@pytest.mark.parametrize("maxc,q,md,bs", [(200, 0.01, 10, 3), (0, 0.007, 10, 10), (5000, 0.02, 3.5, 5),
                                          (1000, 0.05, 0, 3), (0, 0.3, 1, 3), (10000, 0.01, 25, 7)])
def test_good_features_exact(ctx, orc, synth, maxc, q, md, bs):
    img = synth.frame(640, 480, 0, 0, 4321)
    ctx.upload_gray(0, img)
    got = ctx.good_features(0, maxc, q, md, False, bs)
    ref = orc.good_features(img, maxc, q, md, None, bs)
    assert got is not None and ref is not None
    assert got.shape == ref.shape
    assert np.array_equal(got, ref)


def test_pruning_follows_the_survival_rate_and_stays_exact(orc, synth):
    """With maxCorners a real cap only the strongest candidates enter the minDistance stage; how many follows the share
    that survived in the detection before (8x maxCorners at first).  Whatever the history -- the same frame again, a
    much larger minDistance right after a small one (the kept set falls short: the stage is redone on all candidates),
    back again -- the corners are those of the oracle."""
    from iceberg_tracking_code_amd import Context
    w, h = 1000, 800
    img = synth.frame(w, h, 30, -20, 77)
    c = Context(w, h, n_slots=1, max_pts=8192)
    c.upload_gray(0, img)
    seen = []
    for maxc, md in ((400, 5), (400, 5), (400, 5), (400, 60), (400, 60), (150, 5), (400, 5), (2000, 3), (2000, 3)):
        got = c.good_features(0, maxc, 0.005, md, False, 5)
        ref = orc.good_features(img, maxc, 0.005, md, None, 5)
        assert got.shape == ref.shape and np.array_equal(got, ref), (maxc, md)
        seen.append(c.detect_stats())
    c.close()
    # the kept set shrank after the first detection of a kind, and grew back where it had fallen short
    assert seen[1]["candidates"] < seen[0]["candidates"] and seen[2]["candidates"] <= seen[1]["candidates"]
    assert seen[3]["candidates"] > seen[2]["candidates"]


def test_good_features_mask_and_none(ctx, orc, synth):
    img = synth.frame(640, 480, 0, 0, 4321)
    mask = np.zeros_like(img)
    mask[100:300, 200:500] = 255
    ctx.upload_gray(0, img)
    ctx.set_mask(mask)
    got = ctx.good_features(0, 0, 0.01, 10, True, 10)
    ref = orc.good_features(img, 0, 0.01, 10, mask, 10)
    assert np.array_equal(got, ref)
    xs, ys = got[:, 0, 0], got[:, 0, 1]
    assert xs.min() >= 200 and xs.max() < 500 and ys.min() >= 100 and ys.max() < 300
    flat = np.full((480, 640), 9, np.uint8)
    ctx.upload_gray(0, flat)
    assert ctx.good_features(0, 100, 0.01, 10, False, 3) is None
    ctx.set_mask(np.zeros_like(img))
    ctx.upload_gray(0, img)
    assert ctx.good_features(0, 100, 0.01, 10, True, 3) is None
    ctx.set_mask(None)


@pytest.mark.parametrize("win,maxlevel", [((21, 21), 3), ((31, 31), 5), ((35, 35), 4), ((15, 15), 2)])
def test_specialised_equals_generic_kernel(ctx, orc, synth, win, maxlevel):
    """k_lk_fast (one feature per wave, the default), k_lk_multi (several per wave) and k_lk (generic) must agree bit for
    bit, borders included -- also through the fused forward+backward launch, with every criteria form."""
    w, h = 640, 480
    img0, img1 = _pair(synth, w, h, -700, 555)
    rng = np.random.RandomState(win[0])
    pts = _points(rng, 1500, w, h, border=-20.0)
    ctx.upload_gray(0, img0)
    ctx.upload_gray(1, img1)
    a = ctx.pyrlk(0, 1, pts, None, win, maxlevel, CRIT_DEFAULT, 0)
    b = ctx.pyrlk(0, 1, pts, None, win, maxlevel, CRIT_DEFAULT, 0x100)
    c = ctx.pyrlk(0, 1, pts, None, win, maxlevel, CRIT_DEFAULT, 0x200)
    q = orc.pyrlk(img0, img1, pts, None, win, maxlevel, CRIT_DEFAULT)
    for x, y, v, z in zip(a, b, c, q):
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8))
        assert np.array_equal(x.view(np.uint8), v.view(np.uint8))
        assert np.array_equal(x.view(np.uint8), z.view(np.uint8))
    # numbers of points that do not fill the last wave, criteria that end on the count / on a loose or zero epsilon
    from iceberg_tracking_code_amd.context import LK_GENERIC_KERNEL, LK_MULTI_PER_WAVE
    for npts, crit in ((1, CRIT_REF), (2, (1, 4, 0.0)), (3, (2, 0, 0.5)), (5, (3, 7, 0.0)), (1499, (3, 100, 1e-9)),
                       (6, (3, 0, 0.03))):
        got = {}
        for which in (0, LK_GENERIC_KERNEL, LK_MULTI_PER_WAVE):
            ctx.set_lk_kernel(which)
            try:
                got[which] = ctx.track_fb(0, 1, pts[:npts], win, maxlevel, crit)
            finally:
                ctx.set_lk_kernel(0)
        r = orc.track_fb(img0, img1, pts[:npts], win, maxlevel, crit)
        for which, g in got.items():
            for k in ("p1", "p0r", "err_fwd", "err_bwd", "dist", "st_fwd", "st_bwd", "valid"):
                assert np.array_equal(g[k].view(np.uint8), r[k].view(np.uint8)), (which, npts, crit, k)


def test_large_motion_restage(ctx, orc, synth):
    """A shift beyond the staged search margin forces the tile restage path."""
    w, h = 640, 480
    img0, img1 = _pair(synth, w, h, 256 * 9, -256 * 7)
    rng = np.random.RandomState(3)
    pts = _points(rng, 800, w, h, border=30.0)
    ctx.upload_gray(0, img0)
    ctx.upload_gray(1, img1)
    for win, ml in (((21, 21), 0), ((21, 21), 1), ((35, 35), 1)):
        a = ctx.pyrlk(0, 1, pts, None, win, ml, (3, 30, 0.01))
        q = orc.pyrlk(img0, img1, pts, None, win, ml, (3, 30, 0.01))
        for x, z in zip(a, q):
            assert np.array_equal(x.view(np.uint8), z.view(np.uint8))


@pytest.mark.parametrize("bs", [3, 5, 7, 10])
def test_fused_corner_kernel_equals_generic(ctx, orc, synth, bs, monkeypatch):
    """k_eig_nms<BS> (no eigenvalue map in HBM) vs k_min_eig + k_nms_collect vs oracle."""
    img = synth.frame(777, 500, 5, 9, 31)
    mask = np.zeros_like(img)
    mask[3:450, 10:700] = 1
    ctx.upload_gray(0, img)
    ctx.set_mask(mask)
    for use_mask in (False, True):
        fused = ctx.good_features(0, 0, 0.01, 5, use_mask, bs)
        fused_map = ctx.min_eig_map(0, bs)
        monkeypatch.setenv("ICELK_GENERIC_CORNERS", "1")
        generic = ctx.good_features(0, 0, 0.01, 5, use_mask, bs)
        generic_map = ctx.min_eig_map(0, bs)
        monkeypatch.delenv("ICELK_GENERIC_CORNERS")
        ref = orc.good_features(img, 0, 0.01, 5, mask if use_mask else None, bs)
        assert np.array_equal(fused, generic) and np.array_equal(fused, ref)
        assert np.array_equal(fused_map.view(np.uint32), generic_map.view(np.uint32))
    ctx.set_mask(None)


def test_corner_ties_and_plateaus(ctx, orc):
    """Periodic image: many exactly equal responses -> tie-break by raster index must match."""
    yy, xx = np.mgrid[0:240, 0:320]
    img = (((xx // 8) + (yy // 8)) % 2 * 200 + 20).astype(np.uint8)
    ctx.upload_gray(0, img)
    for md in (0, 1, 4, 10):
        got = ctx.good_features(0, 0, 0.05, md, False, 3)
        ref = orc.good_features(img, 0, 0.05, md, None, 3)
        assert (got is None) == (ref is None)
        if got is not None:
            assert np.array_equal(got, ref), md


class _OracleCv:
    """cv2-shaped facade over the oracle, to drive the reference-shaped loop on the CPU."""

    def __init__(self, orc):
        self.o = orc

    def calcOpticalFlowPyrLK(self, a, b, p0, p1, **kw):
        return self.o.pyrlk(a, b, p0, p1, **kw)

    def goodFeaturesToTrack(self, img, mask=None, **kw):
        return self.o.good_features(img, kw["maxCorners"], kw["qualityLevel"], kw["minDistance"], mask,
                                    kw.get("blockSize", 3))


@pytest.mark.parametrize("track_len", [1, 2, 3])
def test_segment_tracker_equals_reference_loop(orc, synth, track_len):
    """Device-resident loop (SegmentTracker) vs the list-of-lists loop of s1:307-450 run on the oracle."""
    from iceberg_tracking_code_amd import SegmentTracker
    from reference_loops import run_reference_loop
    w, h, nfr = 400, 300, 8
    frames, _ = synth.sequence(w, h, nfr, seed=77, max_step_px=2.5)
    mask = np.zeros((h, w), np.uint8)
    mask[20:280, 30:390] = 255
    fp = dict(maxCorners=400, qualityLevel=0.007, minDistance=10, blockSize=10)
    lk = dict(winSize=(35, 35), maxLevel=4, criteria=(3, 25, 0.03))
    ref = run_reference_loop(frames, track_len, fp, lk, mask=mask, cv=_OracleCv(orc))
    trk = SegmentTracker(w, h, track_len, fp, lk, mask=mask, max_pts=4096)
    got = []
    for f in frames:
        seg = trk.push(f)
        if seg is not None:
            got.append(seg)
    trk.close()
    assert len(got) == len(ref) and len(ref) >= 2
    for (gf, gt, gq), (rf, rt, rq) in zip(got, ref):
        rt = np.asarray(rt, np.float32).reshape(len(rt), -1, 2)
        rq = np.asarray(rq, np.float32).reshape(len(rq), -1)
        assert gf == rf
        assert gt.shape == rt.shape and gt.shape[1] == track_len + 1
        assert np.array_equal(gt.view(np.uint32), rt.view(np.uint32))
        assert np.array_equal(gq.view(np.uint32), rq.view(np.uint32))
        assert len(gt) > 100


@pytest.mark.parametrize("maxc", [50, 400, 3000])
def test_topk_pruning_is_invisible(ctx, orc, synth, maxc, monkeypatch):
    """maxCorners-driven candidate pruning (and its unpruned fallback) must not change the corner list."""
    img = synth.frame(900, 700, 3, 4, 11)
    ctx.upload_gray(0, img)
    ref = orc.good_features(img, maxc, 0.007, 10, None, 10)
    pruned = ctx.good_features(0, maxc, 0.007, 10, False, 10)
    monkeypatch.setenv("ICELK_NO_PRUNE", "1")
    plain = ctx.good_features(0, maxc, 0.007, 10, False, 10)
    monkeypatch.delenv("ICELK_NO_PRUNE")
    assert np.array_equal(pruned, ref) and np.array_equal(plain, ref)
    # a mask that leaves fewer corners than 8*maxCorners candidates can supply forces the fallback path
    mask = np.zeros_like(img)
    mask[100:220, 100:260] = 255
    ctx.set_mask(mask)
    got = ctx.good_features(0, maxc, 0.007, 10, True, 10)
    exp = orc.good_features(img, maxc, 0.007, 10, mask, 10)
    assert np.array_equal(got, exp)
    ctx.set_mask(None)


def test_fb_distance_is_numpy_hypot(ctx, orc, synth):
    """s1:329-333 on the device: `dist = np.hypot(abs(p0 - p0r))` in float32 and `valid = dist < 1`, against numpy
    itself -- on distances one ulp either side of 1.0, on pairs where the float32 sqrt form decides differently, and
    on the dist / valid a fused tracker launch returns."""
    from test_oracle_kat import fb_straddle_cases
    from iceberg_tracking_code_amd.context import FB_HYPOT, FB_SQRT
    dx, dy = fb_straddle_cases()
    p0r = np.zeros((len(dx), 2), np.float32)
    p0 = np.stack([dx, dy], 1)
    p0[::2], p0r[::2] = p0r[::2].copy(), p0[::2].copy()
    d = np.abs(p0 - p0r)
    want = np.hypot(d[:, 0], d[:, 1])
    dist, valid = ctx.fb_filter(p0, p0r, 1.0)
    assert np.array_equal(dist.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(valid.astype(bool), want < 1) and 0 < valid.sum() < len(valid)
    ctx.set_fb_distance(FB_SQRT)
    try:
        alt, avalid = ctx.fb_filter(p0, p0r, 1.0)
    finally:
        ctx.set_fb_distance(FB_HYPOT)
    s = (d[:, 0] ** 2 + d[:, 1] ** 2) ** 0.5   # s0_1:99
    assert np.array_equal(alt.view(np.uint32), s.view(np.uint32)) and np.array_equal(avalid.astype(bool), s < 1)
    assert (avalid != valid).sum() > 20
    # the epilogue of the fused launch is the same function of the p0 / p0r it returns
    img0, img1 = _pair(synth, 640, 480, 600, 420)
    pts = orc.good_features(img0, 3000, 0.005, 5, blockSize=5).reshape(-1, 2)
    ctx.upload_gray(0, img0)
    ctx.upload_gray(1, img1)
    g = ctx.track_fb(0, 1, pts, (21, 21), 3, CRIT_DEFAULT)
    dd = np.abs(pts - g["p0r"])
    hyp = np.hypot(dd[:, 0], dd[:, 1])
    assert np.array_equal(g["dist"].view(np.uint32), hyp.view(np.uint32))
    assert np.array_equal(g["valid"].astype(bool), hyp < 1)


@pytest.mark.parametrize("bs", [3, 5, 7, 10])
def test_two_pass_corner_detector(orc, synth, bs, monkeypatch):
    """ICELK_TWO_PASS_CORNERS=1: the integer bracket of the eigenvalue map + OpenCV's float arithmetic at the possible maxima
    only (k_corners_fast.hip) gives the corner list of the one-pass kernel and of the oracle -- with a mask, on a frame
    whose tiles hang over every border, on flat regions (exact zeros), plateaus (ties, list overflow) and frames smaller
    than a window."""
    from iceberg_tracking_code_amd import Context
    monkeypatch.setenv("ICELK_TWO_PASS_CORNERS", "1")
    rng = np.random.RandomState(bs)
    cases = []
    img = synth.frame(701, 467, 5, -3, 77)
    mask = np.zeros_like(img)
    mask[40:400, 60:650] = 255
    mask[100:160, 200:300] = 0
    cases.append((img, None, 0, 0.007, 10))
    cases.append((img, mask, 300, 0.02, 6))
    flat = img.copy()
    flat[150:320, 100:500] = 93                                   # a flat region: eigenvalue exactly zero inside
    cases.append((flat, None, 0, 0.01, 4))
    ramp = (np.add.outer(np.arange(200), 2 * np.arange(300)) % 256).astype(np.uint8)   # plateaus of equal response
    cases.append((ramp, None, 0, 0.05, 0))
    cases.append((rng.randint(0, 256, (9, 13)).astype(np.uint8), None, 0, 0.01, 1))   # smaller than a tile and a window
    cases.append((rng.randint(0, 256, (64, 3)).astype(np.uint8), None, 0, 0.01, 1))
    c = Context(1024, 768, n_slots=1, max_pts=1 << 17)
    for im, m, maxc, q, md in cases:
        c.upload_gray(0, im)
        c.set_mask(m)
        got = c.good_features(0, maxc, q, md, m is not None, bs)
        ref = orc.good_features(im, maxc, q, md, m, bs)
        assert (got is None) == (ref is None), (im.shape, bs)
        if ref is not None:
            assert np.array_equal(got, ref), (im.shape, bs, len(got), len(ref))
    c.set_mask(None)
    # the one-pass kernel on the same handle (switch read per call)
    monkeypatch.delenv("ICELK_TWO_PASS_CORNERS")
    c.upload_gray(0, img)
    assert np.array_equal(c.good_features(0, 0, 0.007, 10, False, bs), orc.good_features(img, 0, 0.007, 10, None, bs))
    c.close()


@pytest.mark.parametrize("name,value", [("lk_sums", 1), ("lk_sums", 2), ("sobel_fma", 1), ("sobel_fma", 2), ("sobel_fma", 3),
                                        ("eig_fma", 1)])
def test_named_variants_equal_the_oracles(orc, synth, name, value):
    """icelk_set_variant: the build-dependent OpenCV semantics SURVEY.md Appendix A asks to keep as named switches -- LK sums
    in the float lanes of the x86 SIMD blocks (3.x / 4.x), Sobel passes and the eigenvalue term fused -- give, bit for
    bit, what the oracle gives under the same switch, and differ from the default somewhere (so the switch is live)."""
    from iceberg_tracking_code_amd import Context
    w, h = 640, 480
    img0, img1 = _pair(synth, w, h, 600, -420, 99)
    c = Context(w, h, n_slots=2, max_pts=1 << 14)
    c.upload_gray(0, img0)
    c.upload_gray(1, img1)
    try:
        if name == "lk_sums":
            pts = orc.good_features(img0, 3000, 0.005, 5, None, 5).reshape(-1, 2)
            base = c.track_fb(0, 1, pts, (21, 21), 3, CRIT_DEFAULT)
            differs = 0
            # the four windows of the tuned kernels (since round 4 the variants run there: k_lk_fast.hip chain_sums) and one
            # that only the window-generic kernel takes
            for win, lvl, crit in (((21, 21), 3, CRIT_DEFAULT), ((35, 35), 4, CRIT_REF), ((31, 31), 5, CRIT_DEFAULT),
                                   ((15, 15), 2, CRIT_REF), ((9, 13), 2, CRIT_DEFAULT)):
                c.set_variant(name, value)
                got = c.track_fb(0, 1, pts, win, lvl, crit)
                c.set_variant(name, 0)
                with orc.variants(**{name: value}):
                    ref = orc.track_fb(img0, img1, pts, win, lvl, crit)
                for k in ("p1", "p0r", "st_fwd", "st_bwd", "err_fwd", "err_bwd", "dist", "valid"):
                    assert np.array_equal(got[k].view(np.uint8), ref[k].view(np.uint8)), (win, k)
                if win == (21, 21):
                    differs = int((got["p1"] != base["p1"]).any(axis=1).sum())
            assert 0 < differs < len(pts) // 2
            again = c.track_fb(0, 1, pts, (21, 21), 3, CRIT_DEFAULT)       # back on the default (tuned kernel)
            assert np.array_equal(again["p1"].view(np.uint32), base["p1"].view(np.uint32))
        else:
            base_map = c.min_eig_map(0, 10)
            c.set_variant(name, value)
            got_map = c.min_eig_map(0, 10)
            got = [c.good_features(0, mc, 0.007, 10, False, bs) for mc, bs in ((0, 10), (500, 3))]
            c.set_variant(name, 0)
            with orc.variants(**{name: value}):
                ref_map = orc.min_eig_map(img0, 10)
                ref = [orc.good_features(img0, mc, 0.007, 10, None, bs) for mc, bs in ((0, 10), (500, 3))]
            assert np.array_equal(got_map.view(np.uint32), ref_map.view(np.uint32))
            assert (got_map != base_map).sum() > 100
            for g, r in zip(got, ref):
                assert np.array_equal(g, r)
            assert np.array_equal(c.min_eig_map(0, 10).view(np.uint32), base_map.view(np.uint32))
    finally:
        c.close()


def test_pyramid_of_forty_random_sizes(orc):
    """The one-launch pyramid reads and writes whole dwords and mirrors level borders inside LDS; what keeps that inside every
    allocation and every region is a property of the slot layout (icelk_abi.hip layout_ok) and of the tile geometry -- not
    of the sizes the other tests happen to use.  Forty random frames from 1 x 1 to 300 x 300 (levels narrower than a halo,
    than a dword, than the filter), both geometries, every level against the oracle.  (Round 2 lost a run to a GPU abort
    in a work-in-progress form of this kernel on the 64 x 48 case -- levels narrower than a halo; that form never reached
    the history.)"""
    from iceberg_tracking_code_amd import Context
    rng = np.random.RandomState(2026)
    sizes = [(1, 1), (2, 2), (3, 1), (1, 5), (4, 4), (5, 300), (300, 5), (64, 48), (65, 129), (128, 128), (129, 127)]
    sizes += [(int(rng.randint(1, 301)), int(rng.randint(1, 301))) for _ in range(29)]
    c = Context(300, 300, n_slots=1, max_pts=64)
    for w, h in sizes:
        img = rng.randint(0, 256, size=(h, w)).astype(np.uint8)
        ref = orc.build_pyramid(img, (3, 3), 8)
        for ahead in (False, True):
            c.upload_gray(0, img)
            if ahead:
                c.build_pyramid_ahead(0, (3, 3), 8)
            assert c.build_pyramid(0, (3, 3), 8) == len(ref) - 1, (w, h)
            for l, r in enumerate(ref):
                assert np.array_equal(c.download_level(0, l), r), (w, h, ahead, l)
    c.close()


@pytest.mark.parametrize("w,h", [(1401, 1150), (2100, 390), (400, 1900)])
def test_pyramid_tile_bands_that_do_not_divide_by_eight(orc, w, h):
    """Round 4: workgroup b of the pyramid launch takes a tile of the band of XCD b % 8 -- the left and right columns of the band
    first, then its inner tiles downwards (upper half of the frame) or upwards (lower half).  Frames whose tile count is no
    multiple of eight (11 x 9 = 99 and 22 x 18 = 396 tiles; 17 x 4; 4 x 15: a short last band, workgroups with no tile),
    bands that begin and end in the middle of a tile row, both geometries, every level against the oracle."""
    from iceberg_tracking_code_amd import Context
    rng = np.random.RandomState(w + h)
    img = rng.randint(0, 256, size=(h, w)).astype(np.uint8)
    ref = orc.build_pyramid(img, (3, 3), 5)
    c = Context(w, h, n_slots=1, max_pts=64)
    try:
        for ahead in (False, True):
            c.upload_gray(0, img)
            if ahead:
                c.build_pyramid_ahead(0, (3, 3), 5)
            assert c.build_pyramid(0, (3, 3), 5) == len(ref) - 1
            for l, r in enumerate(ref):
                assert np.array_equal(c.download_level(0, l), r), (ahead, l)
    finally:
        c.close()


def test_corners_of_forty_random_sizes(orc):
    """The strip kernel walks strips of 245 x 58 outputs (blockSize 10) with per-thread reflected column offsets, reflected
    row loads for the strips at the top and bottom, a register ring that the row loop must meet in phase, and regions of 16
    rows: none of that may depend on the frame sizes the other tests use.  Forty random frames from 3 x 3 to 700 x 500 --
    narrower than a halo, than one strip, a strip and a pixel, a few strips -- with random texture / flat patches, every
    fused blockSize, with and without a random mask, uncapped and capped: the corner list of the oracle, in order."""
    from iceberg_tracking_code_amd import Context
    rng = np.random.RandomState(77)
    sizes = [(3, 3), (4, 7), (11, 9), (58, 58), (59, 60), (245, 58), (246, 59), (247, 117), (490, 116), (491, 175), (64, 48)]
    sizes += [(int(rng.randint(3, 701)), int(rng.randint(3, 501))) for _ in range(29)]
    c = Context(700, 500, n_slots=1, max_pts=1 << 16)
    for k, (w, h) in enumerate(sizes):
        img = rng.randint(0, 256, size=(h, w)).astype(np.uint8)
        if k % 3 == 0:                       # flat patches: plateaus of equal responses, zeros
            y0, x0 = rng.randint(0, h), rng.randint(0, w)
            img[y0:y0 + h // 3 + 1, x0:x0 + w // 3 + 1] = rng.randint(0, 256)
        if k % 4 == 1:                       # smoother texture: fewer, stronger maxima
            img = (img // 32 * 32).astype(np.uint8)
        bs = (3, 5, 7, 10)[k % 4]
        maxc = 0 if k % 2 else int(rng.randint(1, 400))
        md = (1, 3, 10)[k % 3]
        mask = None
        if k % 5 in (1, 3):
            mask = (rng.randint(0, 4, size=(h, w)) > 0).astype(np.uint8) * 255
        c.upload_gray(0, img)
        if mask is not None:
            c.set_mask(mask)
        ref = orc.good_features(img, maxc, 0.01, md, mask, bs)
        got = c.good_features(0, maxc, 0.01, md, mask is not None, bs)
        if ref is None or len(ref) == 0:
            assert got is None or len(got) == 0, (w, h, bs)
        else:
            assert got is not None and np.array_equal(np.asarray(got).reshape(-1, 2), np.asarray(ref).reshape(-1, 2)), (w, h, bs, maxc, md, mask is not None)
    c.close()


def test_device_tail_on_the_detectors_edge_cases(orc):
    """The detections that start a segment take the device-driven tail (k_tail.hip: rank by response bins, maxCorners cut,
    the segment's tables from the device-side counts); icelk_good_features takes the host's (sort after the round trip).
    On the cases where order is delicate -- a periodic frame whose corners all have the SAME response (one response bin
    holds every key: the tie-break by raster address decides), a maxCorners cut through a run of equal responses, masks,
    frames with no corner at all, flat patches, a pruned candidate set that falls short and is redone -- vertex 0 of the
    new segment's tracks is the oracle's corner list, in order."""
    from iceberg_tracking_code_amd import Context
    rng = np.random.RandomState(5)
    c = Context(700, 500, n_slots=1, max_pts=1 << 16)

    def check(img, maxc, q, md, bs, mask=None, what=""):
        c.upload_gray(0, img)
        if mask is not None:
            c.set_mask(mask)
        ref = orc.good_features(img, maxc, q, md, mask, bs)
        n = c.seg_detect(0, maxc, q, md, mask is not None, bs)
        if ref is None or len(ref) == 0:
            assert n == 0, what
            return 0
        tracks, _ = c.seg_read()
        assert n == len(ref) and tracks.shape == (n, 1, 2), what
        assert np.array_equal(tracks[:, 0, :], ref.reshape(-1, 2)), what
        return n

    yy, xx = np.mgrid[0:240, 0:320]
    ties = (((xx // 8) + (yy // 8)) % 2 * 200 + 20).astype(np.uint8)
    d0, h0 = c.seg_tail_stats()
    for md in (1, 4, 10):
        assert check(ties, 0, 0.05, md, 3, what="ties md %d" % md) > 100
    for maxc in (1, 7, 64, 65, 500):          # the cut falls inside the run of equal responses
        assert check(ties, maxc, 0.05, 4, 3, what="ties cut %d" % maxc) == maxc
    assert check(np.full((200, 300), 9, np.uint8), 100, 0.01, 10, 3, what="flat") == 0
    d1, h1 = c.seg_tail_stats()
    assert d1 - d0 == 9 and h1 == h0          # all of them by the device-driven tail
    for k in range(12):
        w, h = int(rng.randint(20, 701)), int(rng.randint(20, 501))
        img = rng.randint(0, 256, size=(h, w)).astype(np.uint8)
        if k % 3 == 0:
            y0, x0 = rng.randint(0, h), rng.randint(0, w)
            img[y0:y0 + h // 3 + 1, x0:x0 + w // 3 + 1] = rng.randint(0, 256)
        mask = (rng.randint(0, 4, size=(h, w)) > 0).astype(np.uint8) * 255 if k % 2 else None
        check(img, 0 if k % 3 else int(rng.randint(1, 400)), 0.01, (1, 3, 10)[k % 3], (3, 5, 7, 10)[k % 4], mask, "random %d" % k)
        c.set_mask(None)
    # a pruned candidate set that falls short of maxCorners: the device verdict hands the tail to the host, which redoes the
    # stage on all candidates
    big = rng.randint(0, 256, size=(480, 640)).astype(np.uint8)
    for maxc, md in ((400, 5), (400, 5), (400, 60), (400, 60), (150, 5)):
        check(big, maxc, 0.005, md, 5, what="pruning history %d %d" % (maxc, md))
    d2, h2 = c.seg_tail_stats()
    assert h2 > h1 and d2 > d1                # both tails were taken
    c.close()


def test_two_pass_detector_with_candidates_prepared_under_another_quality_level(orc, synth, monkeypatch):
    """ICELK_TWO_PASS_CORNERS=1 with icelk_seg_detect_prepare: the candidates of a coming detection are cut at the quality
    level of the detection begun BEFORE (the level of the coming one is not known yet), and icelk_seg_detect_begin adopts
    them only if its own level is not lower.  A lowered, then a raised, then an equal level between
    detections: the corners are the oracle's every time."""
    from iceberg_tracking_code_amd import Context
    monkeypatch.setenv("ICELK_TWO_PASS_CORNERS", "1")
    w, h = 640, 480
    frames, _ = synth.sequence(w, h, 4, seed=12, max_step_px=2.0)
    c = Context(w, h, n_slots=4, max_pts=1 << 15)
    for i, f in enumerate(frames):
        c.upload_gray(i, f)
    prev = None
    for slot, q in ((0, 0.05), (1, 0.005), (2, 0.05), (3, 0.05)):
        c.seg_detect_prepare(slot, False, 10)       # cut at `prev`: higher than q for slot 1 (must not be adopted), lower for slot 2
        n = c.seg_detect(slot, 0, q, 10, False, 10)
        ref = orc.good_features(frames[slot], 0, q, 10, None, 10)
        tracks, _ = c.seg_read()
        assert n == len(ref) > 50 and np.array_equal(tracks[:, 0, :], ref.reshape(-1, 2)), (slot, q, prev)
        prev = q
    c.close()
