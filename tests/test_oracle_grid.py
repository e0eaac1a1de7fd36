"""CPU: oracle/grid_oracle.c against tests/golden/grid_golden.npz -- the grid is the output of the REFERENCE's
create_grid_across_fjord; the per-cell means come from the s3 loop body restated in the generator with the same
matplotlib / numpy calls (see make_grid_golden.py for why the s3 function itself cannot run here)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "grid_golden.npz")


@pytest.fixture(scope="module")
def z():
    return np.load(GOLD, allow_pickle=False)


def grid_of(z):
    """left, top, spacing, cols, rows, cell_on (i * rows + j) rebuilt from the reference's kept indices."""
    left, top = float(min(z["fjord_x"])), float(max(z["fjord_y"]))
    cols, rows = int(z["cols"]), int(z["rows"])
    on = np.zeros(cols * rows, np.uint8)
    on[z["indices"][:, 0] * rows + z["indices"][:, 1]] = 1
    return left, top, float(z["spacing"]), cols, rows, on


def test_grid_cells_match_reference(orc, z):
    """Which cells create_grid_across_fjord keeps = fjord polygon contains the cell centre (tracking_misc.py:49)."""
    left, top, sp, cols, rows, on = grid_of(z)
    centers = np.array([[left + i * sp + 0.5 * sp, top - j * sp - 0.5 * sp] for i in range(cols) for j in range(rows)])
    inside = orc.points_in_polygon(np.stack([z["fjord_x"], z["fjord_y"]], 1), centers)
    assert np.array_equal(inside, on.astype(bool)) and 0 < on.sum() < cols * rows
    kept = centers[inside]
    assert np.array_equal(kept, z["centers"])          # same order (i-major), same floating-point values


def test_binned_means_match_numpy_and_matplotlib(orc, z):
    left, top, sp, cols, rows, on = grid_of(z)
    r = orc.grid_bin(z["px"], z["py"], z["pu"], z["pv"], left, top, sp, cols, rows, on)
    idx = z["indices"][:, 0] * rows + z["indices"][:, 1]
    assert np.array_equal(r["count"][idx], z["counts_all"])
    assert r["count"][on == 0].sum() == 0
    thr = int(z["observation_threshold"])
    meas = idx[z["counts_all"] > thr]
    assert np.array_equal(np.flatnonzero(z["counts_all"] > thr), z["res_grid_id"])
    for key, got in (("u", r["mean_u"]), ("v", r["mean_v"]), ("speed", r["speed"])):
        want = z["res_" + key]
        assert got[meas].tobytes() == want.tobytes(), key
    assert z["res_count"].max() > 1024 and (z["res_count"] < 128).any()
