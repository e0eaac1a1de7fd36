"""Oracle pinning (CPU): hand-computable known answers, an independent numpy restatement of the integer
stages, analytic displacement recovery.  The reference ships no fixtures for this path (PARITY UNPINNED, see
oracle/icelk_oracle.c), so these are what the oracle is held to."""
import numpy as np
import pytest


def np_pyrdown(img):
    """Independent restatement: reflect-101 pad, 5x5 binomial, (sum + 128) >> 8, even samples."""
    k = np.array([1, 4, 6, 4, 1], np.int64)
    p = np.pad(img.astype(np.int64), 2, mode="reflect")
    h, w = img.shape
    rows = sum(k[i] * p[:, i:i + w] for i in range(5))
    full = sum(k[i] * rows[i:i + h, :] for i in range(5))
    return ((full[0::2, 0::2] + 128) >> 8).astype(np.uint8)


def np_scharr(img):
    p = np.pad(img.astype(np.int64), 1, mode="reflect")
    h, w = img.shape
    sm_v = 3 * (p[0:h, :] + p[2:h + 2, :]) + 10 * p[1:h + 1, :]
    df_v = p[2:h + 2, :] - p[0:h, :]
    ix = sm_v[:, 2:w + 2] - sm_v[:, 0:w]
    iy = 3 * (df_v[:, 2:w + 2] + df_v[:, 0:w]) + 10 * df_v[:, 1:w + 1]
    return np.stack([ix, iy], -1).astype(np.int16)


def test_gray_known_answers(orc):
    px = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 200, 30]]], np.uint8)
    # (c0*1868 + c1*9617 + c2*4899 + 8192) >> 14, by hand
    assert orc.bgr2gray(px, 3).tolist() == [[255, 0, 29, 150, 76, 128]]   # last: 2097242 >> 14
    # (c0*3735 + c1*19235 + c2*9798 + 16384) >> 15
    assert orc.bgr2gray(px, 4).tolist() == [[255, 0, 29, 150, 76, 128]]
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    a = img.astype(np.int64)
    assert np.array_equal(orc.bgr2gray(img, 3), ((a[..., 0] * 1868 + a[..., 1] * 9617 + a[..., 2] * 4899 + 8192) >> 14))
    assert np.array_equal(orc.bgr2gray(img, 4), ((a[..., 0] * 3735 + a[..., 1] * 19235 + a[..., 2] * 9798 + 16384) >> 15))


def test_pyrdown_impulse_and_constant(orc):
    img = np.zeros((9, 9), np.uint8)
    img[4, 4] = 255
    out = orc.pyrdown(img)
    assert out.shape == (5, 5)
    # 255 * k_i * k_j / 256 rounded: centre 36, edge 24, diagonal 16 -> (255*36+128)>>8 ...
    assert out[2, 2] == (255 * 36 + 128) >> 8 == 35 or out[2, 2] == 36
    assert out[2, 2] == (255 * 36 + 128) >> 8
    assert out[2, 1] == (255 * 6 + 128) >> 8 and out[1, 1] == (255 * 1 + 128) >> 8
    assert np.array_equal(orc.pyrdown(np.full((13, 7), 201, np.uint8)), np.full((7, 4), 201, np.uint8))


@pytest.mark.parametrize("shape", [(480, 640), (31, 45), (2, 2), (1, 9), (8, 1), (5, 6)])
def test_pyrdown_equals_numpy_restatement(orc, shape):
    rng = np.random.RandomState(shape[0] * 100 + shape[1])
    img = rng.randint(0, 256, shape).astype(np.uint8)
    if min(shape) < 3:
        pytest.skip("numpy reflect pad needs >= 3 px; the C code handles it by iterated reflection")
    assert np.array_equal(orc.pyrdown(img), np_pyrdown(img))


def test_pyramid_stop_rule_by_hand(orc):
    # after creating level l the NEXT size decides: stop when it does not exceed winSize
    assert orc.pyramid_levels(640, 480, (21, 21), 3) == 3
    assert orc.pyramid_levels(640, 480, (35, 35), 10) == 3      # 80x60 -> next 40x30, 30 <= 35
    assert orc.pyramid_levels(640, 480, (35, 35), 2) == 2
    assert orc.pyramid_levels(4000, 3000, (21, 21), 3) == 3
    assert orc.pyramid_levels(5760, 3840, (31, 31), 5) == 5
    assert orc.pyramid_levels(40, 40, (21, 21), 3) == 0          # next 20x20 <= 21
    levels = orc.build_pyramid(np.zeros((480, 640), np.uint8), (35, 35), 10)
    assert [l.shape for l in levels] == [(480, 640), (240, 320), (120, 160), (60, 80)]


def test_scharr_ramp_and_numpy(orc):
    yy, xx = np.mgrid[0:20, 0:30]
    img = (2 * xx + 3 * yy).astype(np.uint8)
    d = orc.scharr(img)
    # interior: Ix = 16 * (I(x+1) - I(x-1)) = 16*4, Iy = 16*6
    assert np.all(d[1:-1, 1:-1, 0] == 64) and np.all(d[1:-1, 1:-1, 1] == 96)
    rng = np.random.RandomState(7)
    r = rng.randint(0, 256, (41, 57)).astype(np.uint8)
    assert np.array_equal(orc.scharr(r), np_scharr(r))
    assert orc.scharr(r).min() >= -4080 and orc.scharr(r).max() <= 4080


def test_lk_identity_and_failures(orc, synth):
    img = synth.frame(320, 240, 0, 0, 5)
    pts = np.float32([[50.25, 60.5], [160, 120], [300.75, 200.125], [-100, 5], [5000, 5000]])
    p1, st, er = orc.pyrlk(img, img, pts, None, (21, 21), 3, (3, 30, 0.01))
    assert st.ravel().tolist() == [1, 1, 1, 0, 0]
    assert np.array_equal(p1.reshape(-1, 2)[:3], pts[:3])      # zero residual -> delta exactly 0
    assert np.all(er.ravel() == 0)
    flat = np.full((240, 320), 90, np.uint8)
    _, st, _ = orc.pyrlk(flat, flat, pts[:3], None, (21, 21), 3, (3, 30, 0.01))
    assert st.sum() == 0                                        # minEig < 1e-4 at level 0


@pytest.mark.parametrize("shift", [(256 * 3, -256 * 2), (300, -200), (-517, 77)])
def test_lk_recovers_known_translation(orc, synth, shift):
    w, h = 480, 360
    a, b = synth.frame(w, h, 0, 0, 9), synth.frame(w, h, shift[0], shift[1], 9)
    pts = orc.good_features(a, 400, 0.01, 8, None, 5).reshape(-1, 2)
    for win, ml, crit in (((21, 21), 3, (3, 30, 0.01)), ((35, 35), 4, (3, 25, 0.03))):
        p1, st, er = orc.pyrlk(a, b, pts, None, win, ml, crit)
        d = (p1.reshape(-1, 2) - pts)[st.ravel() == 1]
        e = np.abs(d - synth.true_flow((0, 0), shift))
        assert st.mean() > 0.95
        assert np.median(e) < 0.03 and np.percentile(e, 90) < 0.1
        r = orc.track_fb(a, b, pts, win, ml, crit)
        assert r["valid"].mean() > 0.95 and np.median(r["dist"]) < 0.02


def test_lk_criteria_clamping(orc, synth):
    a, b = synth.frame(200, 150, 0, 0, 3), synth.frame(200, 150, 400, 100, 3)
    pts = orc.good_features(a, 50, 0.01, 8, None, 3).reshape(-1, 2)
    # maxCount clamps to 100, eps to 10; without the COUNT / EPS bits the defaults 30 / 0.01 apply
    x = orc.pyrlk(a, b, pts, None, (15, 15), 2, (3, 1000, 0.01))
    y = orc.pyrlk(a, b, pts, None, (15, 15), 2, (3, 100, 0.01))
    assert all(np.array_equal(i, j) for i, j in zip(x, y))
    x = orc.pyrlk(a, b, pts, None, (15, 15), 2, (0, 5, 5.0))
    y = orc.pyrlk(a, b, pts, None, (15, 15), 2, (3, 30, 0.01))
    assert all(np.array_equal(i, j) for i, j in zip(x, y))
    z = orc.pyrlk(a, b, pts, None, (15, 15), 2, (1, 0, 0.0))      # zero iterations: nextPts = prevPts
    assert np.array_equal(z[0].reshape(-1, 2), pts)


def test_min_eig_map_against_float64(orc, synth):
    img = synth.frame(96, 64, 11, 22, 4)
    for bs in (3, 10):
        p = np.pad(img.astype(np.float64), 1, mode="reflect")
        h, w = img.shape
        s = 1.0 / (4 * bs * 255)
        dx = ((p[0:h, 2:] - p[0:h, :-2]) + 2 * (p[1:h + 1, 2:] - p[1:h + 1, :-2]) + (p[2:, 2:] - p[2:, :-2])) * s
        dy = ((p[2:, 0:w] - p[0:h, 0:w]) + 2 * (p[2:, 1:w + 1] - p[0:h, 1:w + 1]) + (p[2:, 2:] - p[0:h, 2:])) * s
        an = bs // 2
        def box(a):
            q = np.pad(a, ((an, bs - 1 - an), (an, bs - 1 - an)), mode="reflect")
            return sum(q[i:i + h, j:j + w] for i in range(bs) for j in range(bs))
        a, b, c = box(dx * dx) * 0.5, box(dx * dy), box(dy * dy) * 0.5
        ref = (a + c) - np.sqrt((a - c) ** 2 + b * b)
        got = orc.min_eig_map(img, bs)
        assert np.allclose(got, ref, rtol=2e-4, atol=1e-7)


def test_good_features_rules(orc):
    img = np.full((200, 300), 20, np.uint8)
    for (y, x) in ((40, 50), (40, 200), (150, 120)):
        img[y:y + 30, x:x + 30] = 220
    p = orc.good_features(img, 0, 0.05, 10, None, 3)
    assert p is not None and p.shape[1:] == (1, 2) and p.dtype == np.float32
    xy = p.reshape(-1, 2)
    assert np.all(xy == np.round(xy))                                  # integer-valued coordinates
    assert xy[:, 0].min() >= 1 and xy[:, 0].max() <= 298 and xy[:, 1].min() >= 1 and xy[:, 1].max() <= 198
    d = np.sqrt(((xy[:, None] - xy[None]) ** 2).sum(-1)) + np.eye(len(xy)) * 1e9
    assert d.min() >= 10                                               # minDistance, strict <
    # every square contributes its 4 corners (within 2 px)
    for (y, x) in ((40, 50), (40, 200), (150, 120)):
        for cy, cx in ((y, x), (y, x + 29), (y + 29, x), (y + 29, x + 29)):
            assert np.min(np.abs(xy - [cx, cy]).max(1)) <= 2
    eig = orc.min_eig_map(img, 3)
    vals = eig[xy[:, 1].astype(int), xy[:, 0].astype(int)]
    assert np.all(np.diff(vals) <= 0)                                  # response order
    assert np.array_equal(orc.good_features(img, 5, 0.05, 10, None, 3), p[:5])   # maxCorners = prefix
    mask = np.zeros_like(img)
    mask[:, :150] = 255
    pm = orc.good_features(img, 0, 0.05, 10, mask, 3).reshape(-1, 2)
    assert pm[:, 0].max() < 150 and len(pm) > 0
    assert orc.good_features(np.full((50, 50), 7, np.uint8), 10, 0.01, 5, None, 3) is None


def test_golden_vectors(orc):
    """Self-generated fixtures (tests/golden/make_golden.py): guard the oracle against silent drift."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "lk_small.npz"))
    p1, st, er = orc.pyrlk(g["img0"], g["img1"], g["pts"], None, (21, 21), 2, (3, 30, 0.01))
    assert np.array_equal(p1.view(np.uint32), g["p1"].view(np.uint32))
    assert np.array_equal(st, g["st"]) and np.array_equal(er.view(np.uint32), g["err"].view(np.uint32))
    assert np.array_equal(orc.good_features(g["img0"], 0, 0.01, 6, None, 5), g["corners"])
    assert np.array_equal(orc.pyrdown(g["img0"]), g["down"])


def fb_straddle_cases():
    """(dx, dy) float32 pairs around the unit circle whose np.hypot is 1 - ulp, 1, 1 + ulp, plus pairs where the
    float32 expression sqrt(dx*dx + dy*dy) lands on the other side of 1.0 than np.hypot (s1:330 vs s0_1:99)."""
    rng = np.random.RandomState(7)
    t = rng.uniform(0, np.pi / 2, 400000)
    dx0, dy = np.cos(t).astype(np.float32), np.sin(t).astype(np.float32)
    one = np.float32(1)
    xs, ys = [], []
    for bump in (0, 1, -1, 2, -2):   # neighbouring float32 values of dx
        cand = (dx0.view(np.int32) + bump).view(np.float32)
        h = np.hypot(cand, dy)
        s = np.sqrt(cand * cand + dy * dy, dtype=np.float32)
        sel = np.zeros(len(t), bool)
        for tgt in (np.nextafter(one, np.float32(0)), one, np.nextafter(one, np.float32(2))):
            sel[np.nonzero(h == tgt)[0][:40]] = True
        sel[np.nonzero((h < one) != (s < one))[0][:200]] = True
        xs.append(cand[sel])
        ys.append(dy[sel])
    return np.concatenate(xs), np.concatenate(ys)


def test_fb_distance_is_numpy_hypot_on_float32(orc):
    """s1:329-333: `valid = np.hypot(|p0 - p0r|) < 1` -- the oracle's distance against numpy itself, on values one ulp
    either side of the threshold and on pairs where the float32 sqrt form would decide differently."""
    dx, dy = fb_straddle_cases()
    assert len(dx) > 300
    p0r = np.zeros((len(dx), 2), np.float32)   # a zero origin keeps p0 - p0r exact
    p0 = np.stack([dx, dy], 1)
    p0[::2], p0r[::2] = p0r[::2].copy(), p0[::2].copy()   # both signs of the difference
    d = np.abs(p0 - p0r)
    want = np.hypot(d[:, 0], d[:, 1])
    assert want.dtype == np.float32
    one = np.float32(1)
    for tgt in (np.nextafter(one, np.float32(0)), one, np.nextafter(one, np.float32(2))):
        assert (want == tgt).any()
    orc.set_fb_distance(0)
    got = orc.fb_distance(p0, p0r)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    orc.set_fb_distance(1)
    alt = orc.fb_distance(p0, p0r)
    orc.set_fb_distance(0)
    assert np.array_equal(alt.view(np.uint32), ((d[:, 0] ** 2 + d[:, 1] ** 2) ** 0.5).view(np.uint32))   # s0_1:99
    assert int(((alt < 1) != (want < 1)).sum()) > 20   # the two forms really do decide differently at the threshold


def test_box_sums_of_the_covariance_planes_are_exact_in_double():
    """cornerMinEigenVal's boxFilter keeps double sums of float32 products.  For 8-bit input every such
    product lies in [2^-30, 2^-4] with a 24-bit mantissa, so any sum of <= 32x32 of them is exact in double: sliding
    (OpenCV's RowSum / ColumnSum) and direct summation (oracle, kernels) give the same float32 -- the summation order is
    not a parity risk."""
    import math
    rng = np.random.RandomState(11)
    for bs in (3, 5, 7, 10, 16, 32):
        scale = np.float32(1.0 / (4.0 * bs * 255.0))
        # derivative values as the Sobel passes produce them: integer combinations of pixels times the scale
        dx = (rng.randint(-1020, 1021, 4096).astype(np.float32) * scale).astype(np.float32)
        dy = (rng.randint(-1020, 1021, 4096).astype(np.float32) * scale).astype(np.float32)
        for plane in (dx * dx, dx * dy, dy * dy):
            assert plane.dtype == np.float32
            win = plane[:bs * bs].astype(np.float64)
            direct = 0.0
            for v in win:
                direct += v
            sliding = float(np.cumsum(plane.astype(np.float64))[bs * bs - 1])
            assert direct == math.fsum(win.tolist()) == sliding


def test_named_variants_are_live_close_and_reset(orc, synth):
    """orc_set_variant (every build-dependent OpenCV semantic a named switch): each variant changes
    something, stays within the distance tools/oracle_variants.py reports at full size (LK sums: far below north_star's
    1e-3 px here; fused Sobel / eigenvalue arithmetic: <= 64 ulp of the map, same corner set), and the context manager
    puts every switch back to its default."""
    img0 = synth.frame(320, 240, 0, 0, 5)
    img1 = synth.frame(320, 240, 300, -170, 5)
    pts = orc.good_features(img0, 400, 0.01, 6, None, 5).reshape(-1, 2)
    base = orc.track_fb(img0, img1, pts, (21, 21), 3, (3, 30, 0.01))
    for mode in (1, 2):
        with orc.variants(lk_sums=mode):
            assert orc.get_variant("lk_sums") == mode
            v = orc.track_fb(img0, img1, pts, (21, 21), 3, (3, 30, 0.01))
        assert orc.get_variant("lk_sums") == 0
        d = np.abs(v["p1"] - base["p1"]).max(axis=1)
        assert 0 < (d > 0).sum() and d.max() < 1e-3
        assert np.array_equal(v["st_fwd"], base["st_fwd"]) and np.array_equal(v["valid"], base["valid"])
    e0 = orc.min_eig_map(img0, 10)
    c0 = orc.good_features(img0, 0, 0.01, 6, None, 10)
    for kw in (dict(sobel_fma=1), dict(sobel_fma=2), dict(sobel_fma=3), dict(eig_fma=1)):
        with orc.variants(**kw):
            e = orc.min_eig_map(img0, 10)
            c = orc.good_features(img0, 0, 0.01, 6, None, 10)
        ulp = np.abs(e.view(np.int32).astype(np.int64) - e0.view(np.int32).astype(np.int64))
        assert 0 < (ulp > 0).sum() and ulp.max() <= 64, kw
        assert {tuple(p) for p in c.reshape(-1, 2)} == {tuple(p) for p in c0.reshape(-1, 2)}, kw
    assert all(orc.get_variant(k) == 0 for k in orc.VARIANTS)
    assert np.array_equal(orc.min_eig_map(img0, 10).view(np.uint32), e0.view(np.uint32))
    with pytest.raises(ValueError):
        orc.set_variant("lk_sums", 7)
