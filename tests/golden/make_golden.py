#!/usr/bin/env python3
"""Writes tests/golden/lk_small.npz with the repo's own CPU oracle (NOT the reference: OpenCV, which holds the
reference's arithmetic for this path, is not available, and the reference ships no vectors -- see
oracle/icelk_oracle.c).  The fixture pins the oracle and the HIP path against drift."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from iceberg_tracking_code_amd import synth  # noqa: E402

img0 = synth.frame(160, 120, 0, 0, 2024)
img1 = synth.frame(160, 120, 333, -190, 2024)
rng = np.random.RandomState(1)
pts = np.stack([rng.uniform(-3, 163, 96), rng.uniform(-3, 123, 96)], 1).astype(np.float32)
p1, st, err = oracle.pyrlk(img0, img1, pts, None, (21, 21), 2, (3, 30, 0.01))
corners = oracle.good_features(img0, 0, 0.01, 6, None, 5)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lk_small.npz"), img0=img0, img1=img1,
                    pts=pts, p1=p1, st=st, err=err, corners=corners, down=oracle.pyrdown(img0))
print("wrote lk_small.npz:", len(pts), "points,", len(corners), "corners")
