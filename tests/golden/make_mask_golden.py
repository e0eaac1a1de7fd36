#!/usr/bin/env python3
"""Generates tests/golden/mask_golden.npz by RUNNING THE REFERENCE's Camera.mask_meshgrid (imports/camtools.py:184-211,
matplotlib.path.Path.contains_points underneath) exactly as s1_lucaskanade_tracking.py:285-291 calls it, in the
development container.  Committed: this script and the data (polygons in, masks out); no reference source.

As in make_utm_golden.py, `shapefile` is an empty placeholder (imported at camtools.py:16, used only by the shapefile
reader that fills `maskpoly`), and the Camera instance is created with __new__ with `pic` / `maskpoly` set by hand
(the workbook reader is absent).  Polygons are integer tuples, as x_y_from_shapefile(tuples=1) returns them
(camtools.py:54-61), so many pixel centres lie exactly on edges and vertices: the tie rules are exercised.
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mask_golden.npz")

CASES = [
    # name, (w, h) of the cropped frame, cropleft, croptop, polygon on the UNCROPPED photo
    ("tri", (64, 48), 0, 0, [(5, 5), (60, 10), (20, 44)]),
    ("rect_on_pixels", (40, 30), 0, 0, [(4, 3), (30, 3), (30, 20), (4, 20)]),
    ("concave_crop", (96, 64), 10, 100, [(12, 104), (100, 101), (103, 160), (60, 130), (14, 162), (40, 128)]),
    ("bowtie", (50, 50), 0, 0, [(5, 5), (45, 45), (45, 5), (5, 45)]),
    ("outside_frame", (80, 60), 20, 20, [(0, 0), (140, 30), (70, 130), (10, 70)]),
    ("repeated_vertex_closed", (32, 32), 0, 0, [(2, 2), (28, 2), (28, 2), (28, 28), (2, 28), (2, 2)]),
    ("sliver", (64, 16), 0, 0, [(1, 8), (62, 7), (62, 9)]),
    ("two_vertices", (16, 16), 0, 0, [(2, 2), (12, 12)]),
]


def main():
    sys.path.insert(0, REF)
    sys.modules["shapefile"] = types.ModuleType("shapefile")   # placeholder, see the docstring
    import imports.camtools as ct
    rng = np.random.default_rng(77)
    cases = list(CASES)
    # a fjord-like outline with 60 integer vertices on a 400x300 crop
    ang = np.sort(rng.uniform(0, 2 * np.pi, 60))
    rad = rng.uniform(60, 140, 60)
    poly = [(int(230 + r * np.cos(a) * 1.3), int(1150 + r * np.sin(a))) for a, r in zip(ang, rad)]
    cases.append(("fjord60", (400, 300), 30, 1000, poly))
    out = {"names": np.array([c[0] for c in cases])}
    for name, (w, h), cl, ctop, poly in cases:
        cam = ct.Camera.__new__(ct.Camera)
        cam.pic = dict(cropleft=np.int64(cl), croptop=np.int64(ctop))
        cam.maskpoly = [(int(a), int(b)) for a, b in poly]
        frame_gray = np.zeros((h, w), np.uint8)
        mask = np.zeros_like(frame_gray)
        y, x = np.mgrid[0:frame_gray.shape[0], 0:frame_gray.shape[1]]          # s1:289
        mask1 = cam.mask_meshgrid(x, y, origin="upper left")                   # s1:290
        mask[mask1 == 1] = 255                                                 # s1:291
        out[name + "_poly"] = np.array(cam.maskpoly, np.int64)
        out[name + "_crop"] = np.array([cl, ctop], np.int64)
        out[name + "_mask"] = mask
        print(name, mask.shape, int((mask == 255).sum()))
    import matplotlib
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes; matplotlib", matplotlib.__version__)


if __name__ == "__main__":
    main()
