#!/usr/bin/env python3
"""Host time per SegmentTracker step (wall time inside push_slot), odd (no host sync) and even (detection finish)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iceberg_tracking_code_amd import Context, SegmentTracker, synth  # noqa: E402

w, h, ring, K = 4000, 3000, 12, 400
ctx = Context(w, h, n_slots=ring, max_pts=1 << 14)
sh = synth.shifts(ring, seed=1234)
for i in range(ring):
    ctx.synth_frame(i, w, h, int(sh[i, 0]), int(sh[i, 1]), 1234)
ctx.sync()
fp = dict(maxCorners=10000, qualityLevel=0.007, minDistance=10, blockSize=10)
lk = dict(winSize=(21, 21), maxLevel=3, criteria=(3, 30, 0.01))
trk = SegmentTracker(w, h, 2, feature_params=fp, lk_params=lk, ctx=ctx)
o, i, d = [], 0, 1
for _ in range(K + 4):
    o.append(i)
    if i + d < 0 or i + d >= ring:
        d = -d
    i += d
t = np.zeros(K)
t0 = time.perf_counter()
for k in range(K):
    a = time.perf_counter()
    trk.push_slot(o[k], False, *o[k + 1:k + 7])
    t[k] = time.perf_counter() - a
ctx.sync()
el = time.perf_counter() - t0
print("steps/s %.1f; host us per step: even (detect) mean %.1f median %.1f; odd mean %.1f median %.1f; total host %.1f%% of wall"
      % (K / el, 1e6 * t[20::2].mean(), 1e6 * np.median(t[20::2]), 1e6 * t[21::2].mean(), 1e6 * np.median(t[21::2]),
         100 * t.sum() / el))
trk.close()
