"""GPU tests at BASELINE.json's full sizes (4000x3000, 10k features; 5760x3840, 31x31 / maxLevel 5) through
size-independent properties, plus an oracle comparison on a crop where the oracle finishes in seconds."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H = 4000, 3000
LK_C2 = dict(winSize=(21, 21), maxLevel=3, criteria=(3, 30, 0.01))


@pytest.fixture(scope="module")
def big():
    from iceberg_tracking_code_amd import Context
    c = Context(5760, 3840, n_slots=3, max_pts=1 << 17)
    yield c
    c.close()


def test_golden_fixture_on_gpu(big):
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "lk_small.npz"))
    big.upload_gray(0, g["img0"])
    big.upload_gray(1, g["img1"])
    p1, st, er = big.pyrlk(0, 1, g["pts"], None, (21, 21), 2, (3, 30, 0.01))
    assert np.array_equal(p1.view(np.uint32), g["p1"].view(np.uint32))
    assert np.array_equal(st, g["st"]) and np.array_equal(er.view(np.uint32), g["err"].view(np.uint32))
    assert np.array_equal(big.good_features(0, 0, 0.01, 6, False, 5), g["corners"])
    assert np.array_equal(big.download_level(0, 1), g["down"])


def test_c2_device_frames_equal_numpy_band(big, synth):
    big.synth_frame(0, W, H, 300, -200, 1234)
    full = big.download_level(0, 0)
    assert full.shape == (H, W)
    for y0 in (0, 1499, 2990):
        assert np.array_equal(full[y0:y0 + 10], synth.frame(W, H, 300, -200, 1234, rows=(y0, y0 + 10)))


def test_c2_detect_and_track_properties(big, synth):
    sh = (597, 264)
    big.synth_frame(0, W, H, 0, 0, 1234)
    big.synth_frame(1, W, H, sh[0], sh[1], 1234)
    corners = big.good_features(0, 10000, 0.007, 10, False, 10)
    assert corners is not None and corners.shape == (10000, 1, 2)
    xy = corners.reshape(-1, 2)
    assert np.all(xy == np.round(xy)) and xy.min() >= 1 and xy[:, 0].max() <= W - 2 and xy[:, 1].max() <= H - 2
    # minDistance: no two corners closer than 10 px (grid hashing keeps this O(n))
    cell = {}
    for i, (x, y) in enumerate(xy.astype(int)):
        cell.setdefault((x // 10, y // 10), []).append(i)
    for (cx, cy), idx in cell.items():
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for j in cell.get((cx + dx, cy + dy), []):
                    for i in idx:
                        if i < j:
                            assert (xy[i, 0] - xy[j, 0]) ** 2 + (xy[i, 1] - xy[j, 1]) ** 2 >= 100
    # response order: the eigenvalue map sampled at the corners is non-increasing
    eig = big.min_eig_map(0, 10)
    vals = eig[xy[:, 1].astype(int), xy[:, 0].astype(int)]
    assert np.all(np.diff(vals) <= 0) and vals[-1] > 0.007 * eig.max()
    # the capped list is a prefix of a longer one
    more = big.good_features(0, 12000, 0.007, 10, False, 10)
    assert np.array_equal(more[:10000], corners)
    # tracking: identical frames -> fixed point, zero error; shifted frame -> the known flow, FB distance ~ 0
    same = big.track_fb(0, 0, xy, **LK_C2)
    assert np.array_equal(same["p1"], xy) and same["valid"].all() and np.all(same["err_fwd"] == 0)
    r = big.track_fb(0, 1, xy, **LK_C2)
    ok = r["valid"].astype(bool)
    e = np.abs((r["p1"] - xy)[ok] - synth.true_flow((0, 0), sh))
    assert ok.mean() > 0.98 and np.median(e) < 0.03 and np.percentile(e, 99) < 0.3
    assert np.median(r["dist"][ok]) < 0.02
    # fused call == two plain calls
    p1, st, er = big.pyrlk(0, 1, xy, None, **LK_C2)
    p0r, st2, er2 = big.pyrlk(1, 0, p1, None, **LK_C2)
    assert np.array_equal(p1.reshape(-1, 2).view(np.uint32), r["p1"].view(np.uint32))
    assert np.array_equal(p0r.reshape(-1, 2).view(np.uint32), r["p0r"].view(np.uint32))
    assert np.array_equal(er.ravel().view(np.uint32), r["err_fwd"].view(np.uint32))
    assert np.array_equal(st.ravel(), r["st_fwd"]) and np.array_equal(st2.ravel(), r["st_bwd"])


def test_c2_crop_against_oracle(big, orc, synth):
    """A pyramid level-l pixel depends on level-0 pixels within < 2^(l+2) px, so tracking points well inside a crop
    is the same computation on the crop and on the full frame: compare the GPU (full frame) with the oracle (crop)."""
    sh = (-431, 388)
    big.synth_frame(0, W, H, 0, 0, 77)
    big.synth_frame(1, W, H, sh[0], sh[1], 77)
    x0, y0, cw, ch = 1800, 1200, 640, 480
    a = synth.frame(W, H, 0, 0, 77, rows=(y0, y0 + ch))[:, x0:x0 + cw]
    b = synth.frame(W, H, sh[0], sh[1], 77, rows=(y0, y0 + ch))[:, x0:x0 + cw]
    # x0, y0 are multiples of 2^maxLevel, so the crop's pyramid samples the same grid as the full frame's
    rng = np.random.RandomState(4)
    margin = 150
    local = np.stack([rng.uniform(margin, cw - margin, 500), rng.uniform(margin, ch - margin, 500)], 1).astype(np.float32)
    glob = (local + np.float32([x0, y0])).astype(np.float32)
    got = big.track_fb(0, 1, glob, **LK_C2)
    ref = orc.track_fb(a, b, local, **LK_C2)
    assert np.array_equal(got["st_fwd"], ref["st_fwd"]) and np.array_equal(got["valid"], ref["valid"])
    # not bit-comparable: the same point is x ~ 2000 in one frame and x ~ 300 in the other, and float32 carries
    # 8x fewer sub-pixel bits at 2048 than at 256, so the bilinear weights differ in their last units
    assert np.abs((got["p1"] - np.float32([x0, y0])) - ref["p1"]).max() < 2e-3
    assert np.abs(got["err_fwd"] - ref["err_fwd"]).max() < 0.02
    # eigenvalue map: interior of the crop is bit-identical
    eig_full = big.min_eig_map(0, 10)
    eig_crop = orc.min_eig_map(a, 10)
    assert np.array_equal(eig_full[y0 + 16:y0 + ch - 16, x0 + 16:x0 + cw - 16].view(np.uint32),
                          eig_crop[16:-16, 16:-16].view(np.uint32))


def test_c5_geometry_runs(big, synth):
    """BASELINE.json configs[4]: 5760x3840, 50k features, 31x31, maxLevel 5 (LDS / occupancy stress)."""
    w, h = 5760, 3840
    big.synth_frame(0, w, h, 0, 0, 5)
    big.synth_frame(1, w, h, 512, -256, 5)
    assert big.build_pyramid(0, (31, 31), 5) == 5
    assert [big.download_level(0, l).shape for l in (0, 5)] == [(3840, 5760), (120, 180)]
    corners = big.good_features(0, 50000, 0.007, 8, False, 10)
    assert corners.shape == (50000, 1, 2)
    r = big.track_fb(0, 1, corners, winSize=(31, 31), maxLevel=5, criteria=(3, 30, 0.01))
    ok = r["valid"].astype(bool)
    e = np.abs((r["p1"] - corners.reshape(-1, 2))[ok] - synth.true_flow((0, 0), (512, -256)))
    assert ok.mean() > 0.98 and np.median(e) < 0.03


def test_segment_tracker_c2_counts(synth):
    """Device-resident loop at full size: segment tables have the np.savez shapes and the known motion."""
    from iceberg_tracking_code_amd import SegmentTracker
    fp = dict(maxCorners=10000, qualityLevel=0.007, minDistance=10, blockSize=10)
    trk = SegmentTracker(W, H, 2, fp, LK_C2, max_pts=1 << 14)
    sh = synth.shifts(5, seed=1234)
    segs = []
    for i in range(5):
        s = trk.push_synth(int(sh[i, 0]), int(sh[i, 1]), 1234)
        if s is not None:
            segs.append(s)
    trk.close()
    assert [s[0] for s in segs] == [0, 2]
    for first, tracks, quality in segs:
        assert tracks.shape[1:] == (3, 2) and quality.shape == (len(tracks), 2) and 9000 < len(tracks) <= 10000
        for v in range(2):
            flow = synth.true_flow(sh[first + v], sh[first + v + 1])
            assert np.median(np.abs((tracks[:, v + 1] - tracks[:, v]) - flow)) < 0.03
        assert np.all(quality < 1.0)


def test_c3_sixty_four_pairs_streamed_from_pinned_memory(synth, orc):
    """BASELINE.json configs[2] at its real size: 65 consecutive 4000x3000 frames cross PCIe from pinned host memory
    (hipMemcpyAsync, three uploads in flight, six slots) while the segments are tracked.  Checked through properties that
    do not need a CPU run at this size -- every segment comes back, same tracks as the loop over HBM-resident frames (bit
    for bit), recovered flow = the known motion -- plus the oracle on a crop of one pair."""
    import ctypes as C
    from iceberg_tracking_code_amd import Context, SegmentTracker
    n, T = 65, 2
    fp = dict(maxCorners=10000, qualityLevel=0.007, minDistance=10, blockSize=10)
    sh = synth.shifts(n, seed=77)
    af = synth.affines(n, seed=77)
    res = Context(W, H, n_slots=n, max_pts=1 << 14)
    for i in range(n):
        res.synth_frame(i, W, H, int(sh[i, 0]), int(sh[i, 1]), 77, affine=af[i])
    res.sync()
    # reference run: frames resident, everything done at its own frame (no look-ahead)
    ref = SegmentTracker(W, H, T, fp, LK_C2, ctx=res, lookahead=False)
    want = [s for s in (ref.push_slot(i, wait=True) for i in range(n)) if s is not None]
    assert len(want) == (n - 1) // T
    # the same frames parked in pinned host memory
    pinned = []
    for i in range(n):
        img = np.ascontiguousarray(res.download_level(i, 0))
        p = res.host_alloc(W * H)
        C.memmove(p, img.ctypes.data, W * H)
        pinned.append(p)
        if i == 10:
            crop0 = img[1000:1400, 1500:2100].copy()
        if i == 11:
            crop1 = img[1000:1400, 1500:2100].copy()
    res.close()
    trk = SegmentTracker(W, H, T, fp, LK_C2, max_pts=1 << 14, n_slots=6)
    got = []
    for i in range(3):
        trk.prefetch_pinned(pinned[i], W)
    for i in range(n):
        if i + 3 < n:
            trk.prefetch_pinned(pinned[i + 3], W)
        s = trk.push_prefetched(wait=True)
        if s is not None:
            got.append(s)
    trk.ctx.sync()
    for p in pinned:
        trk.ctx.host_free(p)
    trk.close()
    assert len(got) == len(want) == 32
    for (fa, ta, qa), (fb, tb, qb) in zip(want, got):
        assert fa == fb and ta.shape[1:] == (T + 1, 2) and len(ta) > 8000
        assert np.array_equal(ta, tb) and np.array_equal(qa, qb)
    # the tracks follow the motion the frames were generated with (translation + <= 0.5 % deformation: compare with the
    # exact displacement of the texture point under each track's first vertex)
    first, tracks, _ = got[5]
    one = 1 << 20

    def tex(i, xy):   # texture coordinate sampled at pixel xy of frame i
        x, y = xy[:, 0].astype(np.float64), xy[:, 1].astype(np.float64)
        return np.stack([x + sh[i, 0] / 256.0 + (af[i, 0] * x + af[i, 1] * y) / one,
                         y + sh[i, 1] / 256.0 + (af[i, 2] * x + af[i, 3] * y) / one], 1)
    d = np.abs(tex(first, tracks[:, 0]) - tex(first + 1, tracks[:, 1]))   # the same texture point must be under both
    assert np.median(d) < 0.05 and np.percentile(d, 95) < 0.25
    # oracle on a crop of one pair (frames 10 -> 11), features of the crop detected by the oracle itself
    pts = orc.good_features(crop0, 500, 0.007, 10, None, 10).reshape(-1, 2)
    c = Context(600, 400, n_slots=2, max_pts=4096)
    c.upload_gray(0, crop0)
    c.upload_gray(1, crop1)
    g = c.track_fb(0, 1, pts, **LK_C2)
    c.close()
    r = orc.track_fb(crop0, crop1, pts, **LK_C2)
    for k in ("p1", "p0r", "err_fwd", "err_bwd", "dist", "st_fwd", "st_bwd", "valid"):
        assert np.array_equal(g[k].view(np.uint8), r[k].view(np.uint8)), k


def test_segment_archive_and_iteration_counts(synth):
    """icelk_seg_archive (the device-side read-out of a finished segment that a sharded run gathers over RCCL) gives the
    rows icelk_seg_read gives; icelk_prof_iterations reports one (forward, backward) count per live track."""
    import ctypes as C
    from iceberg_tracking_code_amd import Context
    hip = C.CDLL("libamdhip64.so")
    w, h = 1024, 768
    c = Context(w, h, n_slots=3, max_pts=4096)
    sh = synth.shifts(3, seed=5)
    for i in range(3):
        c.synth_frame(i, w, h, int(sh[i, 0]), int(sh[i, 1]), 5)
    n0 = c.seg_detect(0, 2000, 0.007, 10, False, 10)
    c.prof_enable(True)
    c.seg_track(0, 1, **LK_C2)
    c.seg_track(1, 2, **LK_C2)
    itf, itb = c.prof_iterations()
    c.prof_enable(False)
    tracks, quality = c.seg_read()
    assert 1000 < len(tracks) <= n0 and len(itf) == len(itb) and len(tracks) <= len(itf) <= n0
    assert itf.min() >= 1 and itf.max() <= 4 * 30 and itb.max() <= 4 * 30
    rows = 2048
    bufs = [C.c_void_p() for _ in range(3)]
    for b, nbytes in zip(bufs, (rows * 3 * 2 * 4, rows * 2 * 4, 4)):
        assert hip.hipMalloc(C.byref(b), C.c_size_t(nbytes)) == 0
    nv = c.seg_archive(bufs[0].value, bufs[1].value, bufs[2].value, rows)
    c.sync()
    cnt = np.zeros(1, np.int32)
    at = np.zeros((rows, 3, 2), np.float32)
    aq = np.zeros((rows, 2), np.float32)
    for host, b in ((at, bufs[0]), (aq, bufs[1]), (cnt, bufs[2])):
        assert hip.hipMemcpy(C.c_void_p(host.ctypes.data), b, C.c_size_t(host.nbytes), 2) == 0   # device to host
    assert nv == 3 and cnt[0] == len(tracks)
    assert np.array_equal(at[:cnt[0]], tracks) and np.array_equal(aq[:cnt[0]], quality)
    from iceberg_tracking_code_amd._lib import IcelkError
    with pytest.raises(IcelkError, match="-4"):
        c.seg_archive(bufs[0].value, bufs[1].value, bufs[2].value, 10)   # fewer rows than the segment started with
    # the closed form: after a switch the segment above is the closed one, the new one is current
    with pytest.raises(IcelkError, match="-5"):
        c.seg_archive(bufs[0].value, bufs[1].value, bufs[2].value, rows, closed=True)   # no switch yet: no closed segment
    n1 = c.seg_detect(2, 2000, 0.007, 10, False, 10)
    assert n1 > 1000
    hip.hipMemset(bufs[0], 0, C.c_size_t(rows * 3 * 2 * 4))
    nv = c.seg_archive(bufs[0].value, bufs[1].value, bufs[2].value, rows, closed=True)
    c.sync()
    for host, b in ((at, bufs[0]), (aq, bufs[1]), (cnt, bufs[2])):
        assert hip.hipMemcpy(C.c_void_p(host.ctypes.data), b, C.c_size_t(host.nbytes), 2) == 0
    assert nv == 3 and cnt[0] == len(tracks)
    assert np.array_equal(at[:cnt[0]], tracks) and np.array_equal(aq[:cnt[0]], quality)
    t2, q2 = c.seg_read(closed=True)
    assert np.array_equal(t2, tracks) and np.array_equal(q2, quality)
    t3, _ = c.seg_read()                                  # the current one: just detected, one vertex
    assert t3.shape == (n1, 1, 2)
    for b in bufs:
        hip.hipFree(b)
    c.close()
