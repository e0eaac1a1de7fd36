"""Against tests/golden/ref_functions_golden.npz: outputs of two reference functions RUN AS THEY ARE
(tests/golden/make_ref_functions_golden.py): `Camera.__init__` with tide correction (camtools.py:111-179) and
`utm_to_gridded_utm` (s3_utm_to_gridded_utm.py:222-446).  Round 1 pinned `utm.CameraModel` and the per-cell means
against restatements inside the golden generators; these vectors come from the functions themselves.
CPU part: CameraModel, oracle projection, oracle gridding.  GPU part (marked): the kernels."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_functions_golden.npz")


@pytest.fixture(scope="module")
def z():
    return np.load(GOLD, allow_pickle=False)


def _models(z):
    from iceberg_tracking_code_amd.utm import CameraModel
    cols = [str(c) for c in z["calib_cols"]]
    rows = {"cam1": dict(zip(cols, z["calib_rows"][0])), "cam2": dict(zip(cols, z["calib_rows"][1]))}
    fields = [str(f) for f in z["cam_fields"]]
    out = []
    for name, vals in zip(z["cam_names"], z["cam_values"]):
        r = rows[str(name)]
        want = dict(zip(fields, vals))
        m = CameraModel(int(r["image_width"]), int(r["image_height"]), r["sensor_width"], r["easting"], r["northing"],
                        r["elevation"], r["antenna_height"], r["theta"], r["phi"], r["psi"], r["sigma"],
                        int(r["crop_left"]), int(r["crop_right"]), int(r["crop_top"]), int(r["crop_bottom"]),
                        tide_elevation=want["tide"])
        out.append((m, want))
    return out


def test_camera_model_equals_the_reference_constructor(z):
    """Every entry of the `cam` / `pic` dictionaries Camera.__init__ builds (np.radians of the angles, sigma scaled by
    width / chipsize, height lowered by the tide of the minute), bit for bit."""
    for m, want in _models(z):
        for k in ("chipsize", "E", "N", "H", "theta", "phi", "psi", "sigma"):
            assert np.float64(m.cam[k]).tobytes() == np.float64(want[k]).tobytes(), k
        for k in ("width", "height", "cropleft", "cropright", "croptop", "cropbottom"):
            assert float(m.pic[k]) == want[k], k
    assert len({w["H"] for _, w in _models(z)}) == 3          # three different tides


def test_projection_through_the_constructed_camera_oracle(orc, z):
    """photo_to_utm(photocords_cropped_to_uncropped(1234.5, 321.25)) of the instance the reference's constructor made."""
    for m, want in _models(z):
        tr = np.array([[[1234.5, 321.25], [1240.0, 325.0]]], np.float32)
        r = orc.project_tracks(tr, m.as_dict(), dict(interval_s=60, max_speed=1e9, min_speed=0.0, max_speedfactor=1e9,
                                                     max_angle=180, speed_threshold=1e9))
        assert np.float64(r["x"][0, 0]).tobytes() == np.float64(want["utm_x"]).tobytes()
        assert np.float64(r["y"][0, 0]).tobytes() == np.float64(want["utm_y"]).tobytes()


def _windows(z):
    for fi in range(int(z["grid_n_files"])):
        g = lambda k: z["grid_%02d_%s" % (fi, k)]   # noqa: E731
        yield str(g("name")), g


def test_gridding_oracle_equals_the_reference_function(orc, z):
    """The four 30-minute windows utm_to_gridded_utm wrote for the synthetic day: cell selection, counts, means and speeds
    of oracle/grid_oracle.c on the velocities the function had selected, bit for bit."""
    fx, fy = z["grid_fjord_x"], z["grid_fjord_y"]
    sp, thr = float(z["grid_spacing"]), int(z["grid_threshold"])
    left, top = float(min(fx)), float(max(fy))
    n_win = 0
    for name, g in _windows(z):
        cols, rows = int(g("cols")), int(g("rows"))
        centers = np.array([[left + i * sp + 0.5 * sp, top - j * sp - 0.5 * sp] for i in range(cols) for j in range(rows)])
        on = orc.points_in_polygon(np.stack([fx, fy], 1), centers).astype(np.uint8)
        r = orc.grid_bin(g("px"), g("py"), g("pu"), g("pv"), left, top, sp, cols, rows, on)
        kept = np.flatnonzero(on)                                   # i * rows + j of the kept cells, in the order s3 walks
        meas = kept[r["count"][kept] > thr]
        assert np.array_equal(np.flatnonzero(r["count"][kept] > thr), g("grid_id")), name
        assert np.array_equal(meas // rows, g("i")) and np.array_equal(meas % rows, g("j"))
        assert np.array_equal(r["count"][meas], g("count"))
        for key, got in (("u", r["mean_u"]), ("v", r["mean_v"]), ("speed", r["speed"])):
            assert got[meas].tobytes() == np.asarray(g(key), np.float64).tobytes(), (name, key)
        assert np.array_equal(centers[meas], np.stack([g("x"), g("y")], 1))
        assert np.array_equal(np.asarray(g("topleft"), np.float64), [left + 0.5 * sp, top - 0.5 * sp])
        n_win += 1
    assert n_win == 4


@pytest.mark.gpu
def test_gridding_kernels_equal_the_reference_function(ctx, z):
    from iceberg_tracking_code_amd import bin_velocities
    fjord = {"x": z["grid_fjord_x"], "y": z["grid_fjord_y"]}
    for name, g in _windows(z):
        r = bin_velocities(ctx, g("px"), g("py"), g("pu"), g("pv"), fjord, int(z["grid_spacing"]), int(z["grid_threshold"]))
        for key in ("grid_id", "i", "j", "count"):
            assert np.array_equal(np.array(r[key], np.int64), np.asarray(g(key)).astype(np.int64)), (name, key)
        for key in ("x", "y", "u", "v", "speed"):
            assert np.array(r[key], np.float64).tobytes() == np.asarray(g(key), np.float64).tobytes(), (name, key)


@pytest.mark.gpu
def test_projection_kernel_through_the_constructed_camera(ctx, z):
    from iceberg_tracking_code_amd import project_tracks
    for m, want in _models(z):
        tr = np.array([[[1234.5, 321.25], [1240.0, 325.0]]], np.float32)
        r = project_tracks(ctx, tr, m, 60, max_speed=1e9, min_speed=0.0, max_speedfactor=1e9, max_angle=180,
                           speed_threshold=1e9)
        assert np.float64(r["x"][0, 0]).tobytes() == np.float64(want["utm_x"]).tobytes()
        assert np.float64(r["y"][0, 0]).tobytes() == np.float64(want["utm_y"]).tobytes()
