"""GPU: the projection epilogue (k_project_tracks through icelk_project_tracks / icelk_seg_project and utm.py)
against (a) the golden vectors the reference itself produced and (b) the oracle on larger seeded inputs.
float64, bit-exact: x, y, u, v, speed and the keep decisions (speed = the restated glibc hypot on both sides; the
only libm-dependent value, acos, enters a comparison only -- see DESIGN.md)."""
import os

import numpy as np
import pytest

import utm_golden as G

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def z():
    return G.load()


@pytest.fixture(scope="module")
def uctx():
    from iceberg_tracking_code_amd import Context
    c = Context(64, 64, n_slots=1, max_pts=1 << 17)
    yield c
    c.close()


def test_photo_to_utm_matches_reference(uctx, z):
    from iceberg_tracking_code_amd import project_tracks
    xy = z["p2u_xy"]
    r = project_tracks(uctx, np.stack([xy, xy], 1), G.camera(z, float(z["p2u_tide"])), 60, 1e9, 0, 1e9, 1e9, 1e9)
    assert G.same_bits(np.stack([r["x"][:, 0], r["y"][:, 0]], 1), z["p2u_utm"])
    assert np.all(r["speed"] == 0) and r["keep"].all()


@pytest.mark.parametrize("name", G.SCENARIOS)
def test_cam_to_utm_writes_the_reference_files(uctx, z, name, tmp_path):
    """utm.cam_to_utm on the golden folders: same file names, same arrays (values, dtypes, order) as
    s2_cam_to_utm.cam_to_utm wrote."""
    from iceberg_tracking_code_amd import cam_to_utm
    ins, outs, filt = G.scenario(z, name)
    src, dst = tmp_path / "in", tmp_path / "utm"
    src.mkdir()
    dst.mkdir()
    tides = {}
    for fname, tracks, tide in ins:
        np.savez(src / fname, tracks=tracks)
        tides[fname.split("_")[0]] = tide
    written = cam_to_utm([str(p) for p in src.iterdir()], str(dst), lambda stamp: G.camera(z, tides[stamp]), ctx=uctx,
                         **filt)
    assert [n for n, _ in written] == [n for n, _ in outs] == sorted(os.listdir(dst))
    for (fname, _), (_, want) in zip(written, outs):
        got = np.load(dst / fname, allow_pickle=False)
        for key in ("x", "y", "u", "v", "speed", "time"):
            assert G.same_bits(got[key], want[key]), (fname, key)


def test_single_vector_track_raises_like_the_reference(uctx, z):
    from iceberg_tracking_code_amd import project_tracks
    with pytest.raises(ValueError):
        project_tracks(uctx, z["t1_tracks"], G.camera(z, float(z["t1_tide"])), 60, 1.7, 0.0, 2.5, 60, 0.0001)
    r = project_tracks(uctx, z["t1_tracks"], G.camera(z, float(z["t1_tide"])), 60, 1.7, 0.0, 2.5, 60, 10.0)
    assert r["keep"].tolist() == [True]          # below speed_threshold the pair criteria are never evaluated


def _random_tracks(rng, n, nv, w=3456, h=1300):
    t = np.zeros((n, nv, 2))
    t[:, 0] = np.stack([rng.uniform(0, w, n), rng.uniform(150, h, n)], 1)
    step = rng.normal(0, 1, (n, 1, 2)) * rng.choice([0.0, 0.02, 0.3, 2.0, 30.0], (n, 1, 1))
    jitter = rng.normal(0, 0.05, (n, nv - 1, 2)) * rng.choice([0.0, 1.0, 8.0], (n, 1, 1))
    t[:, 1:] = t[:, :1] + np.cumsum(step + jitter, 1)
    return t.astype(np.float32)


@pytest.mark.parametrize("nv,filt", [
    (2, dict(max_speed=1.7, min_speed=0.0, max_speedfactor=2.5, max_angle=60, speed_threshold=1e9)),
    (3, dict(max_speed=1.7, min_speed=0.0, max_speedfactor=2.5, max_angle=60, speed_threshold=0.1)),
    (5, dict(max_speed=1.2, min_speed=0.02, max_speedfactor=2.0, max_angle=45, speed_threshold=0.05)),
    (9, dict(max_speed=3.0, min_speed=0.01, max_speedfactor=3.0, max_angle=75, speed_threshold=0.08)),
    (10, dict(max_speed=3.0, min_speed=0.01, max_speedfactor=3.0, max_angle=75, speed_threshold=0.08)),
    (17, dict(max_speed=5.0, min_speed=0.03, max_speedfactor=4.0, max_angle=90, speed_threshold=0.02)),
])
def test_large_inputs_equal_oracle(uctx, orc, z, nv, filt):
    """100 000 tracks per case (the reference's segments hold ~1e4): every output bit and every decision."""
    from iceberg_tracking_code_amd import project_tracks
    rng = np.random.default_rng(100 + nv)
    tracks = _random_tracks(rng, 100000, nv)
    cam = G.camera(z, 0.37)
    got = project_tracks(uctx, tracks, cam, 30, **filt)
    want = orc.project_tracks(tracks, cam.as_dict(), dict(interval_s=30, **filt))
    for key in ("x", "y", "u", "v", "speed"):
        assert G.same_bits(got[key], want[key]), key
    assert not np.any(want["keep"] == 2)
    assert np.array_equal(got["keep"], want["keep"] == 1)
    frac = got["keep"].mean()
    assert 0.02 < frac < 0.98          # both outcomes are exercised


def test_degenerate_inputs(uctx, orc, z):
    from iceberg_tracking_code_amd import project_tracks
    cam = G.camera(z, 0.0)
    f = dict(max_speed=1.7, min_speed=0.0, max_speedfactor=2.5, max_angle=60, speed_threshold=0.1)
    r = project_tracks(uctx, np.zeros((0, 3, 2), np.float32), cam, 60, **f)
    assert r["x"].shape == (0, 2) and r["keep"].shape == (0,)
    with pytest.raises(ValueError):
        project_tracks(uctx, np.zeros((4, 1, 2), np.float32), cam, 60, **f)     # no vector at all
    # rays at / above the horizon: division by ~0 and negative ranges must come out as on the CPU
    t = np.array([[[1700.0, -1000.0 + k], [1701.0, -999.5 + k], [1703.0, -999.0 + k]] for k in range(0, 1200, 7)],
                 np.float32)
    got = project_tracks(uctx, t, cam, 60, **f)
    want = orc.project_tracks(t, cam.as_dict(), dict(interval_s=60, **f))
    for key in ("x", "y", "u", "v", "speed"):
        assert G.same_bits(got[key], want[key]), key
    assert np.array_equal(got["keep"], want["keep"] == 1)


def test_segment_projection_equals_host_array_projection(uctx, z, synth):
    """icelk_seg_project (gather + projection on the device) == seg_read followed by icelk_project_tracks."""
    from iceberg_tracking_code_amd import SegmentTracker, project_segment, project_tracks
    w, h = 800, 600
    frames, _ = synth.sequence(w, h, 3, seed=17, max_step_px=2.0)
    trk = SegmentTracker(w, h, 4, dict(maxCorners=3000, qualityLevel=0.007, minDistance=10, blockSize=10),
                         dict(winSize=(21, 21), maxLevel=3, criteria=(3, 30, 0.01)), max_pts=1 << 14)
    for f in frames:
        trk.push(f, wait=False)
    cam = G.camera(z, 0.2)
    a = project_segment(trk.ctx, cam, 60)
    tracks, _ = trk.ctx.seg_read()
    b = project_tracks(trk.ctx, tracks, cam, 60)
    trk.close()
    assert tracks.shape[1] == 3 and len(tracks) > 500
    for key in ("x", "y", "u", "v", "speed", "keep"):
        assert G.same_bits(a[key], b[key]), key
