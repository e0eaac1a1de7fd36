"""GPU: the cv2-shaped module surface (api.py), the reference-shaped loops driven through it, and the pinned
double-buffered streaming ingest (BASELINE.json configs[0] and configs[2])."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class OracleCv:
    def __init__(self, orc):
        self.o = orc

    def calcOpticalFlowPyrLK(self, a, b, p0, p1, **kw):
        return self.o.pyrlk(a, b, p0, p1, **kw)

    def goodFeaturesToTrack(self, img, mask=None, **kw):
        return self.o.good_features(img, kw["maxCorners"], kw["qualityLevel"], kw["minDistance"], mask,
                                    kw.get("blockSize", 3))


def test_cv2_shaped_functions(orc, synth):
    import iceberg_tracking_code_amd as cv2
    rgb = synth.rgb_from_gray_seeded(321, 200, 10, -20, 8)
    gray = cv2.cvtColor(rgb, cv2.COLOR_BGR2GRAY)
    assert gray.dtype == np.uint8 and gray.shape == (200, 321)
    assert np.array_equal(gray, orc.bgr2gray(rgb, 4))   # default = the 4.x coefficients (environment.yml:254)
    # the reference hands PIL's RGB arrays to COLOR_BGR2GRAY (s1:310-311): channel 0 gets the 0.114 weight
    assert np.array_equal(cv2.cvtColor(rgb, cv2.COLOR_RGB2GRAY), orc.bgr2gray(rgb[:, :, ::-1].copy(), 4))
    cv2.set_gray_variant(3)
    assert np.array_equal(cv2.cvtColor(rgb, cv2.COLOR_BGR2GRAY), orc.bgr2gray(rgb, 3))
    cv2.set_gray_variant(4)
    a, b = synth.frame(400, 300, 0, 0, 5), synth.frame(400, 300, 200, 300, 5)
    mask = np.zeros_like(a)
    mask[:, 100:] = 255
    p = cv2.goodFeaturesToTrack(a, mask=mask, **cv2.REF_FEATURE_PARAMS)
    assert np.array_equal(p, orc.good_features(a, 50000000, 0.007, 10, mask, 10)) and p.shape[1:] == (1, 2)
    p1, st, err = cv2.calcOpticalFlowPyrLK(a, b, p, None, **cv2.REF_LK_PARAMS)
    q1, qs, qe = orc.pyrlk(a, b, p, None, **cv2.REF_LK_PARAMS)
    assert p1.shape == (len(p), 1, 2) and st.shape == (len(p), 1) and err.shape == (len(p), 1)
    assert np.array_equal(p1.view(np.uint32), q1.view(np.uint32)) and np.array_equal(st, qs)
    assert np.array_equal(err.view(np.uint32), qe.view(np.uint32))
    assert cv2.goodFeaturesToTrack(np.full((60, 80), 3, np.uint8), 10, 0.01, 5) is None
    cv2.release()


def test_reference_loops_through_the_gpu_module(orc, synth):
    """The s1-shaped loop and the s0_1-shaped class give the same lists on the GPU module and on the oracle."""
    import iceberg_tracking_code_amd as cv2
    frames, _ = synth.sequence(320, 240, 5, seed=3, max_step_px=2.0)
    fp = dict(maxCorners=150, qualityLevel=0.007, minDistance=10, blockSize=10)
    lk = dict(winSize=(35, 35), maxLevel=4, criteria=(3, 25, 0.03))
    from reference_loops import LucasKanade, run_reference_loop
    got = run_reference_loop(frames, 2, fp, lk, cv=cv2.api)
    ref = run_reference_loop(frames, 2, fp, lk, cv=OracleCv(orc))
    assert len(got) == len(ref) == 2
    for (gf, gt, gq), (rf, rt, rq) in zip(got, ref):
        assert gf == rf and np.array_equal(np.float32(gt), np.float32(rt)) and np.array_equal(np.float32(gq), np.float32(rq))
    # BASELINE.json configs[0]: two 640x480 frames, 200 corners, through the s0_1-shaped class
    f2, _ = synth.sequence(640, 480, 2, seed=1234)
    fp200 = dict(maxCorners=200, qualityLevel=0.007, minDistance=10, blockSize=10)
    a = LucasKanade(f2, 3, 120, cv=cv2.api, feature_params=fp200)
    b = LucasKanade(f2, 3, 120, cv=OracleCv(orc), feature_params=fp200)
    ta, tb = a.run(), b.run()
    assert a.track_counts == b.track_counts == [0] and len(ta) == len(tb) > 150
    assert np.array_equal(np.float32(ta), np.float32(tb))
    cv2.release()


def test_pinned_streaming_equals_blocking_upload(synth):
    """icelk_upload_gray_async from pinned memory (copy stream + events) must give the segments of the blocking path."""
    from iceberg_tracking_code_amd import SegmentTracker, _lib
    w, h, n = 512, 384, 7
    frames, _ = synth.sequence(w, h, n, seed=9, max_step_px=2.0)
    fp = dict(maxCorners=500, qualityLevel=0.007, minDistance=10, blockSize=10)
    lk = dict(winSize=(21, 21), maxLevel=3, criteria=(3, 30, 0.01))

    def run(pinned):
        trk = SegmentTracker(w, h, 2, fp, lk, max_pts=4096)
        segs, bufs = [], []
        lib = _lib.load()
        for i, f in enumerate(frames):
            if pinned:
                # two pinned buffers, refilled alternately: the copy of frame i+1 may overlap the kernels of frame i
                if len(bufs) < 2:
                    ptr = C.c_void_p()
                    assert lib.icelk_host_alloc(C.byref(ptr), w * h) == 0
                    bufs.append(ptr)
                ptr = bufs[i % 2]
                trk.ctx.sync()   # the buffer's previous copy has been consumed
                C.memmove(ptr, f.ctypes.data, w * h)
                s = trk.push_pinned(ptr.value, w)
            else:
                s = trk.push(f)
            if s is not None:
                segs.append(s)
        trk.ctx.sync()
        trk.close()
        for ptr in bufs:
            lib.icelk_host_free(ptr)
        return segs

    a, b = run(False), run(True)
    assert len(a) == len(b) == 3
    for (fa, ta, qa), (fb, tb, qb) in zip(a, b):
        assert fa == fb and np.array_equal(ta, tb) and np.array_equal(qa, qb) and len(ta) > 300


@pytest.mark.parametrize("track_len", [1, 2, 3])
def test_prefetched_uploads_equal_blocking_upload(synth, track_len):
    """Frames t+1 and t+2 are put on the copy stream before frame t is tracked (5 slots, per-slot last-use
    events), which also lets the tracker start detection work one and two steps early (seg_detect_prepare /
    seg_detect_begin on the look-ahead frames): the segments must be those of the blocking path."""
    from iceberg_tracking_code_amd import Context, SegmentTracker
    w, h, n = 640, 360, 9
    frames, _ = synth.sequence(w, h, n, seed=21, max_step_px=2.0)
    fp = dict(maxCorners=400, qualityLevel=0.007, minDistance=10, blockSize=10)
    lk = dict(winSize=(21, 21), maxLevel=3, criteria=(3, 30, 0.01))
    ref = SegmentTracker(w, h, track_len, fp, lk, max_pts=4096, lookahead=False)
    want = [s for s in (ref.push(f) for f in frames) if s is not None]
    ref.close()
    trk = SegmentTracker(w, h, track_len, fp, lk, max_pts=4096, n_slots=5)
    ptrs = []
    for f in frames:   # every frame gets its own pinned buffer: no refill hazards in the test itself
        p = trk.ctx.host_alloc(w * h)
        C.memmove(p, f.ctypes.data, w * h)
        ptrs.append(p)
    got = []
    trk.prefetch_pinned(ptrs[0], w)
    trk.prefetch_pinned(ptrs[1], w)
    with pytest.raises(RuntimeError):
        trk.push(frames[0])                     # mixing sources while uploads are pending is refused
    for i in range(n):
        if i + 2 < n:
            trk.prefetch_pinned(ptrs[i + 2], w)
        s = trk.push_prefetched()
        if s is not None:
            got.append(s)
    with pytest.raises(RuntimeError):
        trk.push_prefetched()
    trk.ctx.sync()
    for p in ptrs:
        trk.ctx.host_free(p)
    trk.close()
    assert len(got) == len(want) == (n - 1) // track_len
    for (fa, ta, qa), (fb, tb, qb) in zip(want, got):
        assert fa == fb and np.array_equal(ta, tb) and np.array_equal(qa, qb) and len(ta) > 200
        assert ta.shape[1] == track_len + 1


@pytest.mark.parametrize("track_len", [2, 3, 4])
@pytest.mark.parametrize("depth", [1, 2, 3, 4, 6])
def test_resident_ring_with_lookahead_equals_serial_loop(synth, track_len, depth):
    """Frames resident in HBM (bench.py's source): with the next 1 .. 4 slots known, the work of a coming detection
    frame moves ahead of it (candidates c-4, min-distance c-3, the host round trip + the new segment's initialisation in
    the spare segment set at c-2, only the switch at c).  The segments -- read out at every detection frame -- must be
    those of the loop that does everything at frame c."""
    from iceberg_tracking_code_amd import Context, SegmentTracker
    w, h, n = 640, 360, 14
    frames, _ = synth.sequence(w, h, n, seed=33, max_step_px=2.0)
    fp = dict(maxCorners=300, qualityLevel=0.007, minDistance=10, blockSize=10)
    lk = dict(winSize=(21, 21), maxLevel=3, criteria=(3, 30, 0.01))
    ref = SegmentTracker(w, h, track_len, fp, lk, max_pts=4096, lookahead=False)
    want = [s for s in (ref.push(f) for f in frames) if s is not None]
    ref.close()
    ctx = Context(w, h, n_slots=n, max_pts=4096)
    for i, f in enumerate(frames):
        ctx.upload_gray(i, f)
    trk = SegmentTracker(w, h, track_len, fp, lk, ctx=ctx)
    got = []
    for i in range(n):
        nxt = [i + k if (i + k < n and k <= depth) else None for k in range(1, 7)]
        s = trk.push_slot(i, True, *nxt)
        if s is not None:
            got.append(s)
    n_live, _ = trk.live()
    trk.close()
    assert len(got) == len(want) == (n - 1) // track_len and n_live > 0
    for (fa, ta, qa), (fb, tb, qb) in zip(want, got):
        assert fa == fb and np.array_equal(ta, tb) and np.array_equal(qa, qb) and len(ta) > 150
        assert ta.shape[1] == track_len + 1


def _serial_segments(frames, w, h, track_len, fp, lk):
    from iceberg_tracking_code_amd import SegmentTracker
    ref = SegmentTracker(w, h, track_len, fp, lk, max_pts=4096, lookahead=False)
    want = [s for s in (ref.push(f) for f in frames) if s is not None]
    ref.close()
    return want


@pytest.mark.parametrize("source", ["resident", "push", "prefetch", "prefetch6"])
@pytest.mark.parametrize("track_len", [1, 2, 3])
def test_joint_launch_across_segment_change_equals_serial_loop(synth, track_len, source):
    """With no read-out at the detection frame (wait=False) and the following frame already on the device, the last
    pair of a segment and the first pair of the next go out as ONE tracker launch (icelk_seg_track_defer + switch +
    icelk_seg_track_async in one step); finished segments are reported through on_close and read with the _closed form.
    Segments must be those of the serial loop, whatever the frame source: resident slots, uploads started ahead, and
    plain uploads (where the following frame is not known and every pair has a launch of its own)."""
    from iceberg_tracking_code_amd import Context, SegmentTracker
    w, h, n = 640, 360, 12
    frames, _ = synth.sequence(w, h, n, seed=35, max_step_px=2.0)
    fp = dict(maxCorners=300, qualityLevel=0.007, minDistance=10, blockSize=10)
    lk = dict(winSize=(21, 21), maxLevel=3, criteria=(3, 30, 0.01))
    want = _serial_segments(frames, w, h, track_len, fp, lk)
    ctx = Context(w, h, n_slots={"resident": n, "push": 3, "prefetch": 5, "prefetch6": 9}[source], max_pts=4096)
    trk = SegmentTracker(w, h, track_len, fp, lk, ctx=ctx)
    got = []

    def on_close(first, closed):
        t, q = ctx.seg_read(closed=closed)
        got.append((first, t, q, closed))
    trk.on_close = on_close
    ctx.prof_enable(True)
    if source == "resident":
        for i, f in enumerate(frames):
            ctx.upload_gray(i, f)
        for i in range(n):
            nxt = [i + k if i + k < n else None for k in range(1, 7)]
            assert trk.push_slot(i, False, *nxt) is None
    elif source == "push":
        for f in frames:
            assert trk.push(f, wait=False) is None
    else:
        ptrs = []
        for f in frames:
            p = ctx.host_alloc(w * h)
            C.memmove(p, f.ctypes.data, w * h)
            ptrs.append(p)
        depth = 6 if source == "prefetch6" else 2      # uploads in flight ahead of the frame being tracked
        for i in range(min(depth, n)):
            trk.prefetch_pinned(ptrs[i], w)
        for i in range(n):
            if i + depth < n:
                trk.prefetch_pinned(ptrs[i + depth], w)
            assert trk.push_prefetched(wait=False) is None
    trk.flush()
    ctx.sync()
    prof = ctx.prof_table()
    if source.startswith("prefetch"):
        for p in ptrs:
            ctx.host_free(p)
    trk.close()
    n_pairs = n - 1
    joint = prof.get("lk_fb_pair", {}).get("launches", 0)
    single = prof.get("lk_fb", {}).get("launches", 0)
    assert single + 2 * joint == n_pairs
    if source == "push" or track_len == 1:
        assert joint == 0
    else:
        assert joint >= (n_pairs // track_len) - 2 and any(g[3] for g in got)
    assert len(got) == len(want) == (n - 1) // track_len
    for (fa, ta, qa), (fb, tb, qb, _) in zip(want, got):
        assert fa == fb and np.array_equal(ta, tb) and np.array_equal(qa, qb) and len(ta) > 150
        assert ta.shape[1] == track_len + 1


def test_waiting_pair_goes_out_when_its_result_is_needed(synth):
    """icelk_seg_track_defer launches nothing; icelk_seg_read of that segment, icelk_sync, an upload into one of the
    pair's slots or a second switch launch the waiting pair first.  Each way gives the tracks of icelk_seg_track."""
    from iceberg_tracking_code_amd import Context
    w, h = 640, 360
    frames, _ = synth.sequence(w, h, 4, seed=36, max_step_px=2.0)
    det = (300, 0.007, 10, False, 10)
    lk = ((21, 21), 3, (3, 30, 0.01), 1e-4, 1.0)
    ctx = Context(w, h, n_slots=4, max_pts=4096)
    for i, f in enumerate(frames):
        ctx.upload_gray(i, f)
    ctx.seg_detect(0, *det)
    ctx.seg_track(0, 1, *lk)
    want_t, want_q = ctx.seg_read()
    ctx.prof_enable(True)
    for how in ("read", "sync", "upload", "closed_read", "second_switch"):
        ctx.prof_reset()
        ctx.seg_detect(0, *det)
        ctx.seg_track_defer(0, 1, *lk)
        if how == "sync":
            ctx.sync()
        launches = lambda: ctx.prof_table().get("lk_fb", {}).get("launches", 0)
        assert launches() == (1 if how == "sync" else 0)
        closed = False
        if how == "upload":
            ctx.upload_gray(1, frames[3])       # the pair reads slot 1: it must run before the frame is replaced
            ctx.upload_gray(1, frames[1])
            assert launches() == 1
        elif how in ("closed_read", "second_switch"):
            ctx.seg_detect(1, *det)             # first switch: the pair keeps waiting, its segment is the closed one
            assert launches() == 0
            closed = True
            if how == "second_switch":
                ctx.seg_detect(2, *det)         # no partner came: the pair goes out; its segment is out of reach now
                assert launches() == 1
                continue
        t, q = ctx.seg_read(closed=closed)
        assert launches() == 1
        assert np.array_equal(t, want_t) and np.array_equal(q, want_q) and len(t) > 150
    ctx.close()


def test_two_detections_in_flight(synth):
    """Two detections may be begun before the first is staged (second detector scratch set, third candidate buffer); a
    third _begin, or a one-call detection in between, is refused (ICELK_ESTATE); _stage / _finish take them in the order
    they were begun, and each gives the corners a detection on its own gives -- with and without prepared candidates."""
    from iceberg_tracking_code_amd import Context
    from iceberg_tracking_code_amd._lib import IcelkError
    w, h = 800, 600
    frames, _ = synth.sequence(w, h, 4, seed=41, max_step_px=2.0)
    det = (600, 0.007, 8, False, 10)
    c = Context(w, h, n_slots=4, max_pts=4096)
    for i, f in enumerate(frames):
        c.upload_gray(i, f)
    want = [c.good_features(i, *det) for i in range(4)]
    lk = ((21, 21), 3, (3, 30, 0.01), 1e-4, 1.0)
    for prepared in (False, True):
        if prepared:
            c.seg_detect_prepare(0, False, 10)
            c.seg_detect_prepare(1, False, 10)       # a second buffer: the first stays valid
        c.seg_detect_begin(0, *det)
        c.seg_detect_begin(1, *det)
        with pytest.raises(IcelkError, match="-5"):
            c.seg_detect_begin(2, *det)
        with pytest.raises(IcelkError, match="-5"):
            c.good_features(2, *det)
        c.seg_detect_prepare(2, False, 10)           # the third candidate buffer is free for this
        n0 = c.seg_detect_stage(det[0])
        c.seg_switch()
        t0, _ = c.seg_read()
        c.seg_detect_begin(2, *det)                  # a set is free again; adopts what was prepared
        n1 = c.seg_detect_finish(det[0])             # the oldest in flight: frame 1
        t1, _ = c.seg_read()
        n2 = c.seg_detect_finish(det[0])
        t2, _ = c.seg_read()
        for n, t, ref in ((n0, t0, want[0]), (n1, t1, want[1]), (n2, t2, want[2])):
            assert n == len(ref) > 300 and np.array_equal(t[:, 0, :], ref.reshape(-1, 2))
        assert np.array_equal(c.good_features(3, *det), want[3])     # nothing in flight: the one-call form works again
    c.close()


def test_launch_order_is_invisible(synth, monkeypatch):
    """The spatial launch order of a segment's tracks (k_seg_order + XCD dealing) changes which workgroup tracks
    which feature, never a result: ICELK_NO_ORDER=1 must give identical segments."""
    from iceberg_tracking_code_amd import SegmentTracker
    w, h, n = 1030, 770, 5
    frames, _ = synth.sequence(w, h, n, seed=4, max_step_px=2.5)
    fp = dict(maxCorners=0, qualityLevel=0.007, minDistance=10, blockSize=10)
    lk = dict(winSize=(21, 21), maxLevel=3, criteria=(3, 30, 0.01))

    def run():
        trk = SegmentTracker(w, h, 2, fp, lk, max_pts=1 << 15)
        out = [s for s in (trk.push(f) for f in frames) if s is not None]
        trk.close()
        return out

    a = run()
    monkeypatch.setenv("ICELK_NO_ORDER", "1")
    b = run()
    assert len(a) == len(b) == 2
    for (fa, ta, qa), (fb, tb, qb) in zip(a, b):
        assert fa == fb and np.array_equal(ta, tb) and np.array_equal(qa, qb) and len(ta) > 2000


def test_crop_during_upload_equals_cropping_first(orc, synth):
    """push_bgr(frame, crop=box) (row-pitched upload of the kept region only) == cropping on the host first; the gray
    image is the oracle's cvtColor of the cropped array (s1:310-311 on the box of camtools.py:213-231)."""
    from iceberg_tracking_code_amd import Context
    rgb = synth.rgb_from_gray_seeded(701, 503, 40, -30, 11)
    box = (37, 101, 14, 5)                                        # left, top, right, bottom
    cropped = np.ascontiguousarray(rgb[101:503 - 5, 37:701 - 14])
    c = Context(701, 503, n_slots=2, max_pts=1024)
    c.upload_bgr(0, rgb, 3, crop=box)
    c.upload_bgr(1, cropped, 3)
    a, b = c.download_level(0, 0), c.download_level(1, 0)
    with pytest.raises(ValueError):
        c.upload_bgr(0, rgb, 3, crop=(400, 0, 301, 0))
    c.close()
    assert a.shape == (503 - 106, 701 - 51) and np.array_equal(a, b) and np.array_equal(a, orc.bgr2gray(cropped, 3))


def test_plain_c_caller(tmp_path):
    """tests/abi_smoke.c built with gcc against include/icelk.h + libicelk.so and run as a process of its own: the
    boundary is usable without Python (detect 200 corners, track them forward-backward into a shifted frame)."""
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    libdir = os.path.join(root, "iceberg_tracking_code_amd")
    exe = tmp_path / "abi_smoke"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-I", os.path.join(root, "include"),
                           os.path.join(here, "abi_smoke.c"), "-o", str(exe), "-L", libdir, "-licelk",
                           "-Wl,-rpath," + libdir])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    f = out.stdout.split()
    assert f[0] == "corners" and int(f[1]) == 200 and int(f[3]) > 150
    # the second frame samples the texture 300/256 px further right and 200/256 px further up: content moves the other way
    assert abs(float(f[5]) + 300 / 256.0) < 0.05 and abs(float(f[6]) - 200 / 256.0) < 0.05


def test_sync_covers_the_candidate_stream(orc, synth):
    """icelk_sync waits for all four streams of the handle, the candidate stream of icelk_seg_detect_prepare included:
    after it every ingest path may overwrite the slot, and a later detection of that slot sees the NEW frame (the
    prepared candidates of the old one are dropped)."""
    from iceberg_tracking_code_amd import Context
    hip = C.CDLL("libamdhip64.so")   # the runtime libicelk.so itself runs on: plain device buffers, no torch

    def to_device(a):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(a.nbytes)) == 0
        assert hip.hipMemcpy(p, C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), 1) == 0   # hipMemcpyHostToDevice
        return p

    w, h = 1024, 768
    old = synth.frame(w, h, 0, 0, 21)
    new = [synth.frame(w, h, 0, 0, 30 + k) for k in range(4)]
    rgb = synth.rgb_from_gray_seeded(w, h, 0, 0, 9)
    c = Context(w, h, n_slots=2, max_pts=1 << 16)
    try:
        want = [orc.good_features(f, 500, 0.01, 10, None, 10) for f in new]
        want_rgb = orc.good_features(orc.bgr2gray(rgb, 4), 500, 0.01, 10, None, 10)
        pin = c.host_alloc(w * h)
        dev, dev_rgb = to_device(new[2]), to_device(rgb)
        for path in range(6):
            c.upload_gray(0, old)
            for _ in range(3):
                c.seg_detect_prepare(0, False, 10)     # candidate kernels queued on the fourth stream
                c.upload_gray(1, old)
                c.seg_detect_prepare(1, False, 10)
            c.sync()
            if path == 0:
                c.upload_gray(0, new[0]); ref = want[0]
            elif path == 1:
                C.memmove(pin, new[1].ctypes.data, w * h)
                c.upload_gray_async(0, pin, w, h, w); ref = want[1]
            elif path == 2:
                c.set_gray_device(0, dev.value, w, h, w); ref = want[2]
            elif path == 3:
                c.upload_bgr(0, rgb, 4); ref = want_rgb
            elif path == 4:
                c.cvt_bgr_device(0, dev_rgb.value, w, h, 3 * w, 4); ref = want_rgb
            else:
                c.synth_frame(0, w, h, 0, 0, 33); ref = want[3]
            got = c.good_features(0, 500, 0.01, 10, False, 10)
            assert np.array_equal(got, ref), path
        c.host_free(pin)
        c.sync()
        hip.hipFree(dev)
        hip.hipFree(dev_rgb)
    finally:
        c.close()


def test_uncapped_detection_beyond_max_pts_is_an_error(synth):
    """maxCorners=50_000_000 (s1:240) means "all of them": when more corners pass than the handle was created for, the
    call fails with ICELK_ECAP instead of returning a silently shortened list."""
    from iceberg_tracking_code_amd import Context, REF_FEATURE_PARAMS
    from iceberg_tracking_code_amd._lib import IcelkError
    img = synth.frame(640, 480, 0, 0, 4)
    c = Context(640, 480, n_slots=1, max_pts=64)
    try:
        c.upload_gray(0, img)
        with pytest.raises(IcelkError, match="-4"):
            c.good_features(0, REF_FEATURE_PARAMS["maxCorners"], 0.007, 10, False, 10)
        with pytest.raises(IcelkError, match="-4"):
            c.good_features(0, 0, 0.007, 10, False, 10)
        with pytest.raises(IcelkError, match="-4"):
            c.seg_detect(0, 0, 0.007, 10, False, 10)
        got = c.good_features(0, 64, 0.007, 10, False, 10)   # a real cap within the capacity is fine
        assert got is not None and len(got) == 64
    finally:
        c.close()


def test_abort_after_a_truncated_sequence(synth):
    """A loop that announced frames ahead and stops before they arrive: detections begun / staged ahead
    are in flight and the one-call forms refuse (ICELK_ESTATE); SegmentTracker.abort() abandons them
    (icelk_seg_detect_cancel), counts the frame a joint launch has already tracked as consumed, and both the handle and
    the tracker go on as if nothing had been started ahead: the remaining segments are those of the serial loop."""
    from iceberg_tracking_code_amd import Context, SegmentTracker
    from iceberg_tracking_code_amd._lib import IcelkError
    w, h, n, T = 640, 360, 11, 2
    frames, _ = synth.sequence(w, h, n, seed=37, max_step_px=2.0)
    fp = dict(maxCorners=300, qualityLevel=0.007, minDistance=10, blockSize=10)
    lk = dict(winSize=(21, 21), maxLevel=3, criteria=(3, 30, 0.01))
    want = _serial_segments(frames, w, h, T, fp, lk)
    ctx = Context(w, h, n_slots=n, max_pts=4096)
    for i, f in enumerate(frames):
        ctx.upload_gray(i, f)
    alone = ctx.good_features(9, 300, 0.007, 10, False, 10)
    trk = SegmentTracker(w, h, T, fp, lk, ctx=ctx)
    got = []
    trk.on_close = lambda first, closed: got.append((first,) + ctx.seg_read(closed=closed))
    for i in range(5):
        assert trk.push_slot(i, False, *[i + k if i + k < n else None for k in range(1, 7)]) is None
    with pytest.raises(IcelkError, match="-5"):
        ctx.good_features(9, 300, 0.007, 10, False, 10)          # the detection of frame 8 is in flight
    # a frame pushed into another slot than the one announced for it: its pair has gone out already
    with pytest.raises(RuntimeError, match="announced"):
        trk.push_slot(7, False)
    consumed = trk.abort()
    assert consumed == 6 and trk.cur == 5                       # step 4 sent (3,4) and (4,5) out together
    assert np.array_equal(ctx.good_features(9, 300, 0.007, 10, False, 10), alone)
    trk.on_close = None                                          # from here on segments come back from the push
    for i in range(consumed, n):
        s = trk.push_slot(i, True)
        if s is not None:
            got.append(s)
    trk.close()
    assert [g[0] for g in got] == [w_[0] for w_ in want] == [0, 2, 4, 6, 8]
    for (fa, ta, qa), (fb, tb, qb) in zip(want, got):
        assert np.array_equal(ta, tb) and np.array_equal(qa, qb) and len(ta) > 150


def test_slot_overwritten_while_its_pyramid_is_still_being_built(orc, synth):
    """icelk_build_pyramid_ahead followed at once by a new frame into the same slot (no tracker launch in between): the
    build that is still running on the pyramid stream must not write levels of the OLD frame over the new pyramid (the
    ingest waits for it)."""
    from iceberg_tracking_code_amd import Context
    w, h = 2400, 1800
    a = synth.frame(w, h, 0, 0, 3)
    b = synth.frame(w, h, 900, -700, 4)
    ref = orc.build_pyramid(b, (21, 21), 4)
    c = Context(w, h, n_slots=1, max_pts=64)
    for _ in range(3):
        c.upload_gray(0, a)
        c.build_pyramid_ahead(0, (21, 21), 4)
        c.seg_detect_prepare(0, False, 10)                    # a reader on the candidates stream as well
        c.upload_gray(0, b)
        assert c.build_pyramid(0, (21, 21), 4) == len(ref) - 1
        for l, r in enumerate(ref):
            assert np.array_equal(c.download_level(0, l), r), l
    c.close()


def test_templates_travel_only_along_a_real_sequence(synth, monkeypatch):
    """The backward pass of a pair leaves its templates for the forward pass of the next pair (icelk_seg_track_len_hint) --
    but only a pair whose FIRST frame is the very frame they were built on may take them.  An adversarial order of pairs
    (the same pair again, a jump, a re-uploaded slot, a pair with other LK parameters, a real continuation) gives,
    vertex by vertex, the tracks of a handle that never reuses anything; the counters show where the reuse engaged."""
    from iceberg_tracking_code_amd import Context
    w, h = 960, 540
    frames, _ = synth.sequence(w, h, 7, seed=11, max_step_px=2.0)
    lk = dict(winSize=(21, 21), maxLevel=3, criteria=(3, 30, 0.01))
    lk31 = dict(winSize=(31, 31), maxLevel=3, criteria=(3, 30, 0.01))

    def run(ctx):
        for i, f in enumerate(frames):
            ctx.upload_gray(i, f)
        n = ctx.seg_detect(0, 800, 0.007, 10, False, 10)
        ctx.seg_track(0, 1, **lk)        # leaves templates of frame 1
        ctx.seg_track(1, 2, **lk)        # a real continuation: takes them
        ctx.seg_track(1, 2, **lk)        # the same pair again: the templates at hand are frame 2's -> builds its own
        ctx.seg_track(4, 5, **lk)        # a jump: frame 4 is not the frame the last pair ended on
        ctx.upload_gray(5, frames[6])    # the slot the templates were built on gets another frame
        ctx.seg_track(5, 6, **lk)        # same slot, other content -> builds its own
        ctx.seg_track(6, 3, **lk31)      # the frame is right, the window is not
        ctx.seg_track(3, 2, **lk31)      # a real continuation at 31x31: takes them
        tracks, quality = ctx.seg_read()
        return n, tracks, quality, ctx.seg_template_stats()

    a = Context(w, h, n_slots=7, max_pts=4096)
    na, ta, qa, (taken, left) = run(a)
    a.close()
    monkeypatch.setenv("ICELK_NO_TEMPLATE_REUSE", "1")
    b = Context(w, h, n_slots=7, max_pts=4096)
    nb, tb, qb, off = run(b)
    b.close()
    assert na == nb and na > 300 and len(ta) > 100
    assert np.array_equal(ta, tb) and np.array_equal(qa, qb)
    assert (taken, left) == (2, 7) and off == (0, 0)


@pytest.mark.parametrize("n_old,n_new", [(90, 3000), (3000, 90), (700, 700), (33, 2049), (2600, 2100)])
def test_joint_launch_of_unequal_segments_equals_two_launches(synth, n_old, n_new):
    """The two jobs of a joint tracker launch are dealt to the workgroups in alternating blocks of 32 groups, the rest of
    the shorter job group by group, what is left of the longer one at the end (k_lk_fast.hip): for segments of very
    different sizes -- fewer than one block, one block and a bit, either job the longer one -- both segments come out as
    from two launches of their own."""
    from iceberg_tracking_code_amd import Context
    w, h = 1600, 1200
    frames, _ = synth.sequence(w, h, 4, seed=5, max_step_px=2.0)
    lk = dict(winSize=(21, 21), maxLevel=3, criteria=(3, 30, 0.01))

    def run(joint):
        c = Context(w, h, n_slots=4, max_pts=4096)
        for i, f in enumerate(frames):
            c.upload_gray(i, f)
        assert c.seg_detect(0, n_old, 0.007, 4, False, 10) == n_old
        c.seg_track(0, 1, wait=False, **lk)
        c.seg_detect_begin(2, n_new, 0.007, 4, False, 10)
        assert c.seg_detect_stage(n_new) == n_new
        if joint:
            c.seg_track_defer(1, 2, **lk)
            c.seg_switch()
            c.prof_reset(); c.prof_enable(True)
            c.seg_track(2, 3, wait=False, **lk)
            c.sync(); c.prof_enable(False)
            assert "lk_fb_pair" in c.prof_table()
        else:
            c.seg_track(1, 2, wait=False, **lk)
            c.seg_switch()
            c.seg_track(2, 3, wait=False, **lk)
        old = c.seg_read(closed=True)
        new = c.seg_read()
        c.close()
        return old, new

    (ta, qa), (tb, qb) = run(True)
    (tc, qc), (td, qd) = run(False)
    assert np.array_equal(ta, tc) and np.array_equal(qa, qc) and np.array_equal(tb, td) and np.array_equal(qb, qd)
    assert len(ta) > n_old // 2 and len(tb) > n_new // 2 and ta.shape[1] == 3 and tb.shape[1] == 2


@pytest.mark.parametrize("win,levels", [((15, 15), 2), ((21, 21), 4), ((31, 31), 3)])
def test_template_reuse_changes_nothing(synth, monkeypatch, win, levels):
    """Segments of three pairs on a small frame (features whose windows hang over the frame edge, pyramid levels narrower
    than a tile) with every window that takes part in the template hand-over: tracks and qualities with the hand-over
    equal those without it (ICELK_NO_TEMPLATE_REUSE), and the hand-over did take place."""
    from iceberg_tracking_code_amd import Context, SegmentTracker
    w, h, n, T = 400, 300, 10, 3
    frames, _ = synth.sequence(w, h, n, seed=23, max_step_px=2.5)
    fp = dict(maxCorners=0, qualityLevel=0.005, minDistance=4, blockSize=5)
    lk = dict(winSize=win, maxLevel=levels, criteria=(3, 30, 0.01))

    def run():
        ctx = Context(w, h, n_slots=n, max_pts=8192)
        for i, f in enumerate(frames):
            ctx.upload_gray(i, f)
        trk = SegmentTracker(w, h, T, fp, lk, ctx=ctx)
        segs = []
        trk.on_close = lambda first, closed: segs.append((first,) + ctx.seg_read(closed=closed))
        for i in range(n):
            trk.push_slot(i, False, *[i + k if i + k < n else None for k in range(1, 7)])
        trk.flush()
        ctx.sync()
        st = ctx.seg_template_stats()
        trk.close()
        return segs, st

    a, (taken, left) = run()
    monkeypatch.setenv("ICELK_NO_TEMPLATE_REUSE", "1")
    b, off = run()
    assert len(a) == len(b) >= 2 and off == (0, 0) and taken >= 4 and left >= taken
    for (fa, ta, qa), (fb, tb, qb) in zip(a, b):
        assert fa == fb and np.array_equal(ta, tb) and np.array_equal(qa, qb) and len(ta) > 50


@pytest.mark.parametrize("max_corners", [0, 700])
def test_device_driven_detection_tail_equals_the_hosts(synth, monkeypatch, max_corners):
    """The tail of a detection that starts a segment (sort of the accepted corners, maxCorners cut, the segment's tables =
    the reset of s1:440-448, launch order) runs on the device from the device-side counts (k_tail.hip); the host adopts
    its verdict.  Same segments as with the host's tail (ICELK_HOST_TAIL=1: round 3's path) and as with the host's tail
    BEHIND a device verdict of "not valid" (ICELK_TAIL_FORCE_STATUS: the path a non-converged min-distance relaxation or
    a pruned candidate set that fell short takes), with frames resident and the detector's work spread over the steps
    before its frame; the statistics say which tail staged the segments."""
    from iceberg_tracking_code_amd import Context, SegmentTracker
    w, h, n = 1030, 770, 11
    frames, _ = synth.sequence(w, h, n, seed=17, max_step_px=2.0)
    fp = dict(maxCorners=max_corners, qualityLevel=0.007, minDistance=10, blockSize=10)
    lk = dict(winSize=(21, 21), maxLevel=3, criteria=(3, 30, 0.01))

    def run():
        ctx = Context(w, h, n_slots=n, max_pts=1 << 15)
        for i, f in enumerate(frames):
            ctx.upload_gray(i, f)
        trk = SegmentTracker(w, h, 2, fp, lk, ctx=ctx)
        out = []
        for i in range(n):
            s = trk.push_slot(i, i % 4 == 0, *[i + k if i + k < n else None for k in range(1, 7)])
            if s is not None:
                out.append(s)
        trk.flush()
        tracks, quality = ctx.seg_read()
        stats = ctx.seg_tail_stats()
        trk.close()
        return out, tracks, quality, stats

    dev, dt, dq, dstat = run()
    monkeypatch.setenv("ICELK_HOST_TAIL", "1")
    host, ht, hq, hstat = run()
    monkeypatch.delenv("ICELK_HOST_TAIL")
    monkeypatch.setenv("ICELK_TAIL_FORCE_STATUS", "1")
    forced, ft, fq, fstat = run()
    assert dstat[0] >= 5 and dstat[1] == 0 and hstat[0] == 0 and hstat[1] == dstat[0] and fstat == hstat
    assert len(dev) == len(host) == len(forced) == 2
    for other in (host, forced):
        for (fa, ta, qa), (fb, tb, qb) in zip(dev, other):
            assert fa == fb and np.array_equal(ta, tb) and np.array_equal(qa, qb)
            assert len(ta) > (500 if max_corners else 2000)
    assert np.array_equal(dt, ht) and np.array_equal(dq, hq) and np.array_equal(dt, ft) and np.array_equal(dq, fq)


@pytest.mark.parametrize("budget_mb", [None, 64])
def test_template_tables_survive_a_segment_that_outgrows_the_one_before(synth, monkeypatch, budget_mb):
    """The last pair of a ~2 000-track segment waits (icelk_seg_track_defer) while the next segment, with more than 16 384
    tracks, is staged and switched to; the joint launch then carries both.  Round 3 grew the template tables at that
    moment -- hipFree + hipMalloc under the waiting pair's pointers.  Now they are laid out once, for max_pts rows within
    the budget (or fewer: with 64 MB the big segment does not fit and builds its own templates); either way tracks and
    qualities are those of a handle that never reuses a template."""
    from iceberg_tracking_code_amd import Context
    w, h = 1600, 1200
    frames, _ = synth.sequence(w, h, 4, seed=29, max_step_px=2.0)
    lk = dict(winSize=(21, 21), maxLevel=3, criteria=(3, 30, 0.01))
    small = (2000, 0.007, 10, False, 10)
    big = (30000, 0.0005, 3, False, 3)

    def run():
        ctx = Context(w, h, n_slots=4, max_pts=1 << 16)
        for i, f in enumerate(frames):
            ctx.upload_gray(i, f)
        ctx.seg_track_len_hint(2)
        n_small = ctx.seg_detect(0, *small)
        ctx.seg_track(0, 1, wait=False, **lk)          # leaves templates
        ctx.seg_detect_begin(2, *big)
        ctx.seg_track_defer(1, 2, **lk)                # the last pair of the small segment waits ...
        n_big = ctx.seg_detect_stage(big[0])
        ctx.seg_switch()
        ctx.seg_track(2, 3, wait=False, **lk)          # ... and goes out with the first pair of the big one
        ta, qa = ctx.seg_read(closed=True)
        tb, qb = ctx.seg_read()
        info = ctx.seg_template_info()
        stats = ctx.seg_template_stats()
        ctx.close()
        return n_small, n_big, ta, qa, tb, qb, info, stats

    if budget_mb is not None:
        monkeypatch.setenv("ICELK_TEMPLATE_BUDGET_MB", str(budget_mb))
    a = run()
    monkeypatch.setenv("ICELK_NO_TEMPLATE_REUSE", "1")
    b = run()
    assert a[0] == b[0] == 2000 and a[1] == b[1] == 30000
    for k in (2, 3, 4, 5):
        assert np.array_equal(a[k], b[k])
    assert len(a[2]) > 1000 and len(a[4]) > 10000 and a[2].shape[1] == 3 and a[4].shape[1] == 2
    by, rows, state = a[6]
    assert state == 0 and b[6][2] == 1 and a[7][0] >= 1       # the small segment's second pair took templates
    if budget_mb is None:
        assert rows == 1 << 16
    else:
        assert 0 < rows < a[1] and by <= budget_mb << 19
