"""CPU: oracle/utm_oracle.c against the golden vectors the REFERENCE produced (tests/golden/make_utm_golden.py ran
s2_cam_to_utm.cam_to_utm and Camera.photo_to_utm from /root/reference) -- this is what pins the projection
epilogue.  Bit-exact float64."""
import datetime as dt

import numpy as np
import pytest

import utm_golden as G


@pytest.fixture(scope="module")
def z():
    return G.load()


def _hourly_from_oracle(orc, z, name):
    """The hour bookkeeping of s2_cam_to_utm.py:199-241,293-311,349-363 around oracle.project_tracks."""
    ins, outs, filt = G.scenario(z, name)
    interval = int(ins[0][0].split("_")[-2].split("sec")[0])
    keys = ("x", "y", "u", "v", "speed", "time")
    cur, nxt, written = {k: [] for k in keys}, {k: [] for k in keys}, []
    next_hour = t = None
    for c, (fname, tracks, tide) in enumerate(ins):
        t = dt.datetime.strptime(fname.split("_")[0], "%Y%m%d-%H%M%S")
        if c == 0:
            next_hour = (t + dt.timedelta(hours=1)).hour
        if t.hour == next_hour:
            written.append(((t - dt.timedelta(hours=1)), cur))
            cur, nxt = nxt, {k: [] for k in keys}
            next_hour = (t + dt.timedelta(hours=1)).hour
        cam = G.camera(z, tide).as_dict()
        r = orc.project_tracks(tracks, cam, dict(interval_s=interval, **filt))
        assert not np.any(r["keep"] == 2)
        for i in np.flatnonzero(r["keep"] == 1):
            for k in range(tracks.shape[1] - 1):
                tv = t + dt.timedelta(seconds=k * interval)
                dst = cur if tv.hour == t.hour else nxt
                for key in ("x", "y", "u", "v", "speed"):
                    dst[key].append(r[key][i, k])
                dst["time"].append(int((tv - dt.datetime(1970, 1, 1)).total_seconds()))
    written.append((t, cur))
    return [("{}_{}00_{}s_utm.npz".format(lab.strftime("%Y%m%d"), lab.strftime("%H"), interval),
             {k: np.array(v) for k, v in d.items()}) for lab, d in written], outs


def test_photo_to_utm_matches_reference(orc, z):
    cam = G.camera(z, float(z["p2u_tide"])).as_dict()
    xy = z["p2u_xy"]
    tracks = np.stack([xy, xy], 1)                      # two identical vertices: x, y of vector 0 = the projection
    r = orc.project_tracks(tracks, cam, dict(interval_s=60, max_speed=1e9, min_speed=0, max_speedfactor=1e9,
                                             max_angle=1e9, speed_threshold=1e9))
    got = np.stack([r["x"][:, 0], r["y"][:, 0]], 1)
    assert G.same_bits(got, z["p2u_utm"])
    assert np.all(r["speed"] == 0) and np.all(r["keep"] == 1)


@pytest.mark.parametrize("name", G.SCENARIOS)
def test_hourly_files_match_reference(orc, z, name):
    got, want = _hourly_from_oracle(orc, z, name)
    assert [n for n, _ in got] == [n for n, _ in want]
    for (_, g), (_, w) in zip(got, want):
        assert len(w["x"]) > 100
        for key in ("x", "y", "u", "v", "speed", "time"):
            assert G.same_bits(g[key], w[key]), key


def test_single_vector_track_raises_like_the_reference(orc, z):
    assert str(z["t1_raises"]) == "ValueError"
    cam = G.camera(z, float(z["t1_tide"])).as_dict()
    r = orc.project_tracks(z["t1_tracks"], cam, dict(interval_s=60, max_speed=1.7, min_speed=0.0, max_speedfactor=2.5,
                                                     max_angle=60, speed_threshold=0.0001))
    assert r["keep"].tolist() == [2]
