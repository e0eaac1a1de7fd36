"""Bit-for-bit comparison with the CPU oracle on the FULL frames of BASELINE.json's configurations -- no crops, no
tolerances.  C2 = 4000x3000 / 10 000 corners / 21x21 / maxLevel 3 (configs[1], the headline), C5 = 5760x3840 / 50 000
corners / 31x31 / maxLevel 5 (configs[4]), REF = the reference's own literals (s1_lucaskanade_tracking.py:240-248:
35x35, maxLevel 4, criteria (25, 0.03), maxCorners uncapped) on its typical 3456x2304 frame.

Frames are generated on the device (translation + <= 0.5 % affine deformation, the bench's motion), downloaded, and
handed to the oracle as they are: the oracle detects on the whole frame and tracks EVERY corner, border features
included, so the 83 000-candidate detection with survival-rate pruning, the border-first / XCD-dealt launch order and
the joint tracker launches are all checked against the oracle at the sizes they are benchmarked at.  Wall time of the
oracle part on the GPU box's 16 threads: see the printed lines (`-s`), a few seconds per configuration."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DET = dict(qualityLevel=0.007, minDistance=10, blockSize=10)     # s1:241-243
CFG = {
    "C2": dict(w=4000, h=3000, maxCorners=10000, lk=dict(winSize=(21, 21), maxLevel=3, criteria=(3, 30, 0.01)), seed=1234),
    "C5": dict(w=5760, h=3840, maxCorners=50000, lk=dict(winSize=(31, 31), maxLevel=5, criteria=(3, 30, 0.01)), seed=55),
    "REF": dict(w=3456, h=2304, maxCorners=0, lk=dict(winSize=(35, 35), maxLevel=4, criteria=(3, 25, 0.03)), seed=91),
}
N_FRAMES, T = 5, 2


@pytest.fixture(scope="module", params=list(CFG))
def case(request, synth):
    """A resident ring of five device-generated frames of the configuration + their host copies."""
    from iceberg_tracking_code_amd import Context
    name = request.param
    c = CFG[name]
    sh = synth.shifts(N_FRAMES, seed=c["seed"])
    af = synth.affines(N_FRAMES, seed=c["seed"])
    ctx = Context(c["w"], c["h"], n_slots=N_FRAMES, max_pts=1 << 17)
    for i in range(N_FRAMES):
        ctx.synth_frame(i, c["w"], c["h"], int(sh[i, 0]), int(sh[i, 1]), c["seed"], affine=af[i])
    ctx.sync()
    frames = [ctx.download_level(i, 0) for i in range(N_FRAMES)]
    yield name, c, ctx, frames
    ctx.close()


def _same_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def test_full_frame_detection_is_the_oracles(case, orc):
    """goodFeaturesToTrack (s1:437) on the whole frame: every corner, in order."""
    name, c, ctx, frames = case
    t0 = time.perf_counter()
    ref = orc.good_features(frames[0], c["maxCorners"], DET["qualityLevel"], DET["minDistance"], None, DET["blockSize"])
    t_orc = time.perf_counter() - t0
    got = ctx.good_features(0, c["maxCorners"], DET["qualityLevel"], DET["minDistance"], False, DET["blockSize"])
    print("\n%s: oracle detection %.2f s, %d corners" % (name, t_orc, len(ref)))
    assert ref is not None and got is not None
    assert len(ref) == (c["maxCorners"] or len(ref)) and len(ref) >= 10000
    assert _same_bits(got, ref)
    # a second frame, detected right after (pruning now follows the survival rate of the detection before)
    ref3 = orc.good_features(frames[3], c["maxCorners"], DET["qualityLevel"], DET["minDistance"], None, DET["blockSize"])
    assert _same_bits(ctx.good_features(3, c["maxCorners"], DET["qualityLevel"], DET["minDistance"], False, DET["blockSize"]), ref3)


def test_full_frame_forward_backward_tracking_is_the_oracles(case, orc):
    """calcOpticalFlowPyrLK x 2 + the FB rule (s1:323-333) for every detected corner plus features planted along the
    frame edge (window partly outside, bounds tests, status 0): every output bit for bit."""
    name, c, ctx, frames = case
    w, h = c["w"], c["h"]
    pts = orc.good_features(frames[0], c["maxCorners"], DET["qualityLevel"], DET["minDistance"], None,
                            DET["blockSize"]).reshape(-1, 2)
    rng = np.random.RandomState(7)
    n_edge = 600
    edge = np.stack([rng.uniform(-3, w + 3, n_edge), rng.uniform(-3, h + 3, n_edge)], 1)
    side = rng.randint(0, 4, n_edge)
    edge[side == 0, 0] = rng.uniform(-4, 25, (side == 0).sum())
    edge[side == 1, 0] = rng.uniform(w - 25, w + 4, (side == 1).sum())
    edge[side == 2, 1] = rng.uniform(-4, 25, (side == 2).sum())
    edge[side == 3, 1] = rng.uniform(h - 25, h + 4, (side == 3).sum())
    allp = np.concatenate([pts, edge.astype(np.float32)]).astype(np.float32)
    near = ((pts[:, 0] < 80) | (pts[:, 0] > w - 80) | (pts[:, 1] < 80) | (pts[:, 1] > h - 80)).sum()
    t0 = time.perf_counter()
    ref = orc.track_fb(frames[0], frames[1], allp, **c["lk"])
    t_orc = time.perf_counter() - t0
    got = ctx.track_fb(0, 1, allp, **c["lk"])
    print("\n%s: oracle forward+backward %.2f s for %d features (%d detected within 80 px of the edge + %d planted)"
          % (name, t_orc, len(allp), near, n_edge))
    assert len(allp) >= 2000 + n_edge
    for k in ("p1", "p0r", "st_fwd", "st_bwd", "err_fwd", "err_bwd", "dist", "valid"):
        assert _same_bits(got[k], ref[k]), k
    assert 0 < (ref["st_fwd"] == 0).sum() < n_edge + len(pts) // 10 and ref["valid"].mean() > 0.9
    # the plain calls (s1:323, s1:326) as well
    p1, st, er = ctx.pyrlk(0, 1, allp, None, **c["lk"])
    r1, rs, re = orc.pyrlk(frames[0], frames[1], allp, None, **c["lk"])
    assert _same_bits(p1, r1) and _same_bits(st, rs) and _same_bits(er, re)


def test_full_frame_segments_are_the_reference_loops(case, orc):
    """The device-resident loop with look-ahead (detections prepared / begun / staged frames ahead) and joint tracker
    launches across the segment change, against the list-of-lists loop of s1:307-450 run on the oracle: the np.savez
    payload of both finished segments, bit for bit."""
    from iceberg_tracking_code_amd import SegmentTracker
    from reference_loops import OracleCv, run_reference_loop
    name, c, ctx, frames = case
    fp = dict(maxCorners=c["maxCorners"] if c["maxCorners"] > 0 else 50000000, **DET)
    t0 = time.perf_counter()
    ref = run_reference_loop(frames, T, fp, c["lk"], cv=OracleCv(orc))
    t_orc = time.perf_counter() - t0
    trk = SegmentTracker(c["w"], c["h"], T, dict(maxCorners=c["maxCorners"], **DET), c["lk"], ctx=ctx)
    got = []

    def on_close(first, closed):
        t, q = ctx.seg_read(closed=closed)
        got.append((first, t, q, closed))
    trk.on_close = on_close
    ctx.prof_reset()
    ctx.prof_enable(True)
    for i in range(N_FRAMES):
        nxt = [i + k if i + k < N_FRAMES else None for k in range(1, 7)]
        assert trk.push_slot(i, False, *nxt) is None
    trk.flush()
    ctx.sync()
    prof = ctx.prof_table()
    ctx.prof_enable(False)
    print("\n%s: reference loop on the oracle %.2f s; segments of %s tracks" % (name, t_orc, [len(s[1]) for s in ref]))
    assert len(ref) == len(got) == (N_FRAMES - 1) // T
    assert prof.get("lk_fb_pair", {}).get("launches", 0) >= 1 and any(g[3] for g in got)
    for (rf, rt, rq), (gf, gt, gq, _) in zip(ref, got):
        rt = np.asarray(rt, np.float32).reshape(len(rt), -1, 2)
        rq = np.asarray(rq, np.float32).reshape(len(rq), -1)
        assert rf == gf and gt.shape[1] == T + 1 and len(gt) > 5000
        assert _same_bits(gt, rt) and _same_bits(gq, rq)
