"""GPU: gridding (k_grid.hip through icelk_points_in_polygon / icelk_grid_bin and gridding.py) against
tests/golden/grid_golden.npz (grid from the reference's create_grid_across_fjord; per-cell means from the s3 loop body
restated with matplotlib / numpy in the generator) and against the oracle on a larger seeded set.  float64, bit-exact."""
import numpy as np
import pytest

from test_oracle_grid import GOLD, grid_of

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def z():
    return np.load(GOLD, allow_pickle=False)


def test_grid_equals_reference(ctx, z):
    from iceberg_tracking_code_amd import create_grid_across_fjord
    fjord = {"x": z["fjord_x"], "y": z["fjord_y"]}
    polygons, centers, indices, topleft_c, rows, cols = create_grid_across_fjord(ctx, fjord, int(z["spacing"]))
    assert rows == int(z["rows"]) and cols == int(z["cols"])
    assert np.array_equal(np.array(topleft_c, np.float64), z["topleft_center"])
    assert np.array_equal(np.array(polygons, np.float64), z["polygons"])
    assert np.array_equal(np.array(centers, np.float64), z["centers"])
    assert np.array_equal(np.array(indices, np.int64), z["indices"])


def test_binned_velocities_equal_golden(ctx, z):
    from iceberg_tracking_code_amd import bin_velocities
    fjord = {"x": z["fjord_x"], "y": z["fjord_y"]}
    r = bin_velocities(ctx, z["px"], z["py"], z["pu"], z["pv"], fjord, int(z["spacing"]),
                       int(z["observation_threshold"]))
    assert np.array_equal(r["counts_all"], z["counts_all"])
    for key in ("grid_id", "i", "j", "count"):
        assert np.array_equal(np.array(r[key], np.int64), z["res_" + key].astype(np.int64)), key
    for key in ("x", "y", "u", "v", "speed"):
        assert np.array(r[key], np.float64).tobytes() == z["res_" + key].tobytes(), key
    assert len(r["measured"]) + len(r["not_measured"]) == len(z["polygons"])


def test_large_set_equals_oracle(ctx, orc, z):
    """10^6 velocities, a quarter of them exactly on cell edges or corners, cells with up to ~10^5 observations."""
    left, top, sp, cols, rows, on = grid_of(z)
    rng = np.random.default_rng(5)
    n = 1000000
    x = rng.uniform(left - 100, left + cols * sp + 100, n)
    y = rng.uniform(top - rows * sp - 100, top + 100, n)
    k = n // 8
    x[:k] = left + sp * rng.integers(0, cols + 1, k)
    y[k:2 * k] = top - sp * rng.integers(0, rows + 1, k)
    x[2 * k:3 * k] = x[3 * k:4 * k] * 0 + left + sp * 5.5 + rng.normal(0, 40, k)     # a crowded spot
    y[2 * k:3 * k] = top - sp * 7.5 + rng.normal(0, 40, k)
    u = rng.normal(0, 1, n) * 10.0 ** rng.integers(-4, 3, n)
    v = rng.normal(0, 1, n) * 10.0 ** rng.integers(-4, 3, n)
    from iceberg_tracking_code_amd import _lib
    import ctypes as C
    cnt = np.zeros(cols * rows, np.int32)
    mu, mv, spd = (np.zeros(cols * rows, np.float64) for _ in range(3))
    f = lambda a: a.ctypes.data_as(_lib.f64p)   # noqa: E731
    ctx._ck(ctx._lib.icelk_grid_bin(ctx._h, f(x), f(y), f(u), f(v), n, left, top, sp, cols, rows,
                                    on.ctypes.data_as(_lib.u8p), cnt.ctypes.data_as(_lib.i32p), f(mu), f(mv), f(spd)))
    want = orc.grid_bin(x, y, u, v, left, top, sp, cols, rows, on)
    assert np.array_equal(cnt, want["count"]) and cnt.max() > 50000 and (cnt[on == 1] > 0).all()
    assert mu.tobytes() == want["mean_u"].tobytes() and mv.tobytes() == want["mean_v"].tobytes()
    assert spd.tobytes() == want["speed"].tobytes()
    assert C.sizeof(C.c_double) == 8


def test_points_in_polygon_degenerate(ctx, orc):
    from iceberg_tracking_code_amd import points_in_polygon
    pts = np.array([[0.5, 0.5], [2.0, 2.0], [0.0, 0.0], [1.0, 0.5]])
    sq = [(0, 0), (1, 0), (1, 1), (0, 1)]
    assert np.array_equal(points_in_polygon(ctx, sq, pts), orc.points_in_polygon(sq, pts))
    assert not points_in_polygon(ctx, [(0, 0), (1, 1)], pts).any()
    assert points_in_polygon(ctx, sq, np.zeros((0, 2))).shape == (0,)
