#!/usr/bin/env python3
"""Generates tests/golden/grid_golden.npz (development container only).

(a) RUNS THE REFERENCE: `imports.tracking_misc.create_grid_across_fjord` (tracking_misc.py:23-56), imported from
    /root/reference, on a synthetic fjord outline -> polygons, centre points, indices, topleft, rows, cols.
(b) The loop body of s3_utm_to_gridded_utm.py:391-421 cannot be run as it stands (it sits inside a function that walks
    day folders, reads the calibration workbook and camera files), so its few lines are restated HERE with the very
    third-party calls it makes -- matplotlib.path.Path(poly).contains_points(points), np.sum(...) / n, np.hypot --
    on synthetic velocities that include positions exactly on cell edges and corners.  This pins the primitives'
    semantics (tie rule, pairwise summation order), not the reference's own lines; tests say so.
Committed: this script and the data; no reference source.
"""
import os
import sys

import matplotlib.path as mplPath
import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "grid_golden.npz")


def main():
    sys.path.insert(0, REF)
    import imports.tracking_misc as trm
    rng = np.random.default_rng(99)
    ang = np.sort(rng.uniform(0, 2 * np.pi, 40))
    rad = rng.uniform(900, 2100, 40)
    fjord = {"x": 497000.0 + np.round(1.6 * rad * np.cos(ang), 1), "y": 6521000.0 + np.round(rad * np.sin(ang), 1)}
    spacing = 250
    polygons, centers, indices, topleft_c, rows, cols = trm.create_grid_across_fjord(fjord, spacing)
    out = dict(fjord_x=fjord["x"], fjord_y=fjord["y"], spacing=np.array(spacing), rows=np.array(rows),
               cols=np.array(cols), topleft_center=np.array(topleft_c, np.float64),
               polygons=np.array(polygons, np.float64), centers=np.array(centers, np.float64),
               indices=np.array(indices, np.int64))
    # velocities: a cloud over the fjord + points exactly on cell edges / corners + a heavy cell (> 128 and > 1024
    # observations: every branch of numpy's pairwise sum)
    left, top = min(fjord["x"]), max(fjord["y"])
    n = 20000
    x = rng.uniform(left - 300, left + cols * spacing + 300, n)
    y = rng.uniform(top - rows * spacing - 300, top + 300, n)
    k = 1000
    x[:k] = left + spacing * rng.integers(0, cols + 1, k)                     # on vertical edges
    y[k:2 * k] = top - spacing * rng.integers(0, rows + 1, k)                 # on horizontal edges
    x[2 * k:3 * k] = left + spacing * rng.integers(0, cols + 1, k)            # on corners
    y[2 * k:3 * k] = top - spacing * rng.integers(0, rows + 1, k)
    ci, cj = indices[len(indices) // 2]
    x[3 * k:3 * k + 3000] = rng.uniform(left + ci * spacing, left + (ci + 1) * spacing, 3000)
    y[3 * k:3 * k + 3000] = rng.uniform(top - (cj + 1) * spacing, top - cj * spacing, 3000)
    u = rng.normal(0.1, 0.3, n) * 10.0 ** rng.integers(-3, 2, n)
    v = rng.normal(-0.05, 0.2, n) * 10.0 ** rng.integers(-3, 2, n)
    points = np.vstack((x, y)).T
    disp = np.vstack((u, v)).T
    observation_threshold = 5
    res = {key: [] for key in ("grid_id", "i", "j", "x", "y", "u", "v", "speed", "count")}
    counts_all = []
    for counter, (poly, center, index) in enumerate(zip(polygons, centers, indices)):      # s3:391
        grid = mplPath.Path(poly).contains_points(points)                                   # s3:394
        sel = disp[grid == 1]                                                               # s3:396
        nobs = len(sel)
        counts_all.append(nobs)
        if nobs > observation_threshold:                                                    # s3:400
            mean_u = np.sum(sel[:, 0]) / nobs                                               # s3:408
            mean_v = np.sum(sel[:, 1]) / nobs
            res["grid_id"].append(counter)
            res["i"].append(index[0]); res["j"].append(index[1])
            res["x"].append(center[0]); res["y"].append(center[1])
            res["u"].append(mean_u); res["v"].append(mean_v)
            res["speed"].append(np.hypot(mean_u, mean_v))                                   # s3:413
            res["count"].append(nobs)
    out.update(px=x, py=y, pu=u, pv=v, observation_threshold=np.array(observation_threshold),
               counts_all=np.array(counts_all, np.int64))
    for key, val in res.items():
        out["res_" + key] = np.array(val)
    np.savez_compressed(OUT, **out)
    import matplotlib
    print("cells kept", len(polygons), "of", rows * cols, "measured", len(res["count"]), "max count", max(counts_all),
          "points counted", sum(counts_all), "| wrote", OUT, os.path.getsize(OUT), "bytes; numpy", np.__version__,
          "matplotlib", matplotlib.__version__)


if __name__ == "__main__":
    main()
