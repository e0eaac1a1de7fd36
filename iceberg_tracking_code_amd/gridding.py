"""Gridding of the projected velocities -- host-side mirror of the parts of s3_utm_to_gridded_utm.py that touch every
point: `trm.create_grid_across_fjord` (imports/tracking_misc.py:23-56) and the per-cell selection / averaging loop
(s3:391-421).  Point-in-polygon tests, the per-cell sums (numpy's pairwise order) and the speeds run on the GPU
(`icelk_points_in_polygon`, `icelk_grid_bin`, csrc/k_grid.hip); what stays here is the cell geometry of a few hundred
squares and the packing of the result arrays the reference hands to np.savez (s3:441-445).  No CPU fallback.

Out of scope: the day / camera / time-window bookkeeping around the loop (s3:120-388) and plotting.
"""
import ctypes as C
import math

import numpy as np

from . import _lib


def _f64(a):
    return a.ctypes.data_as(_lib.f64p)


def points_in_polygon(ctx, poly, points):
    """matplotlib.path.Path(poly).contains_points(points) (radius 0) -> bool array."""
    p = np.ascontiguousarray(poly, dtype=np.float64).reshape(-1, 2)
    q = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 2)
    out = np.zeros(len(q), np.uint8)
    ctx._ck(ctx._lib.icelk_points_in_polygon(ctx._h, _f64(p), len(p), _f64(q), len(q), out.ctypes.data_as(_lib.u8p)))
    return out.astype(bool)


def create_grid_across_fjord(ctx, fjord, spacing):
    """What trm.create_grid_across_fjord returns (tracking_misc.py:23-56): [polygons, centerpoints, indices,
    topleft_px_center, rows, cols] -- squares of `spacing` laid from the top-left corner of the fjord outline, column
    by column, kept when the outline contains the cell centre.  All cells are formed at once (the same float64
    expressions per cell: corner = left + i * spacing, top - j * spacing; centre = corner +/- 0.5 * spacing) and the
    containment test runs on the GPU for all centres together.  `fjord` has 'x' and 'y' arrays."""
    fx, fy = np.asarray(fjord["x"]), np.asarray(fjord["y"])
    left, right, bottom, top = min(fx), max(fx), min(fy), max(fy)
    cols = int(math.ceil((right - left) / spacing))
    rows = int(math.ceil((top - bottom) / spacing))
    ii, jj = (a.ravel() for a in np.meshgrid(np.arange(cols), np.arange(rows), indexing="ij"))   # i-major, as s3 walks
    ox, oy = left + ii * spacing, top - jj * spacing
    centers = np.stack([ox + 0.5 * spacing, oy - 0.5 * spacing], 1)
    keep = points_in_polygon(ctx, np.vstack((fx, fy)).T, centers) if len(centers) else np.zeros(0, bool)
    polygons = [[(x, y), (x + spacing, y), (x + spacing, y - spacing), (x, y - spacing)]
                for x, y in zip(ox[keep], oy[keep])]
    return [polygons, [list(c) for c in centers[keep]], [[int(i), int(j)] for i, j in zip(ii[keep], jj[keep])],
            [left + 0.5 * spacing, top - 0.5 * spacing], rows, cols]


def bin_velocities(ctx, x, y, u, v, fjord, spacing, observation_threshold, grid=None):
    """The loop of s3:391-421 for one time window.  Returns the dict the reference saves (s3:441-445, without
    `grid_size` / `topleft` / `rows` / `cols`, which the caller has): grid_id, i, j, x, y, u, v, speed, count,
    measured, not_measured."""
    if grid is None:
        grid = create_grid_across_fjord(ctx, fjord, spacing)
    polygons, centers, indices, _, rows, cols = grid
    fx, fy = np.asarray(fjord["x"]), np.asarray(fjord["y"])
    left, top = float(min(fx)), float(max(fy))
    on = np.zeros(cols * rows, np.uint8)
    idx = np.array([i * rows + j for i, j in indices], np.int64)
    on[idx] = 1
    a = [np.ascontiguousarray(t, dtype=np.float64).ravel() for t in (x, y, u, v)]
    n = len(a[0])
    cnt = np.zeros(cols * rows, np.int32)
    mu, mv, sp = (np.zeros(cols * rows, np.float64) for _ in range(3))
    ctx._ck(ctx._lib.icelk_grid_bin(ctx._h, _f64(a[0]), _f64(a[1]), _f64(a[2]), _f64(a[3]), n, left, top,
                                    float(spacing), cols, rows, on.ctypes.data_as(_lib.u8p),
                                    cnt.ctypes.data_as(_lib.i32p), _f64(mu), _f64(mv), _f64(sp)))
    out = {k: [] for k in ("grid_id", "i", "j", "x", "y", "u", "v", "speed", "count", "measured", "not_measured")}
    for counter, (poly, center, index, c) in enumerate(zip(polygons, centers, indices, idx)):
        nobs = int(cnt[c])
        if nobs > observation_threshold:                                   # s3:400
            out["grid_id"].append(counter)
            out["i"].append(index[0])
            out["j"].append(index[1])
            out["x"].append(center[0])
            out["y"].append(center[1])
            out["u"].append(mu[c])
            out["v"].append(mv[c])
            out["speed"].append(sp[c])
            out["count"].append(nobs)
            out["measured"].append(poly)
        else:
            out["not_measured"].append(poly)
    out["counts_all"] = cnt[idx].astype(np.int64)
    return out
