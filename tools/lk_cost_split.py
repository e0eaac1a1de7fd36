#!/usr/bin/env python3
"""Splits the tracker launch time into its per-level fixed part (tile loads, template, matrix) and its per-iteration
part: the fused forward+backward launch over 20 000 features with the iteration count forced to 1, 2, 3, 5 per level
(criteria = COUNT only).  argv: win maxLevel"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iceberg_tracking_code_amd import Context, synth  # noqa: E402

w, h = 4000, 3000
win = (int(sys.argv[1]), int(sys.argv[1])) if len(sys.argv) > 1 else (21, 21)
maxlevel = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = 20000
ctx = Context(w, h, n_slots=2, max_pts=1 << 16)
sh = synth.shifts(3, seed=1234)
ctx.synth_frame(0, w, h, int(sh[1, 0]), int(sh[1, 1]), 1234)
ctx.synth_frame(1, w, h, int(sh[2, 0]), int(sh[2, 1]), 1234)
pts = ctx.good_features(0, n, 0.007, 10, False, 10).reshape(-1, 2)
n = len(pts)
ctx.track_fb(0, 1, pts[:1000], win, maxlevel)
rows = []
for cnt in (1, 2, 3, 5, 8):
    ctx.prof_reset()
    ctx.prof_enable(True)
    for _ in range(6):
        ctx.track_fb(0, 1, pts, win, maxlevel, criteria=(1, cnt, 0.0))     # TERM_CRITERIA_COUNT only
    ctx.prof_enable(False)
    t = ctx.prof_table()["lk_fb"]["avg_us"]
    rows.append((cnt, t))
    print("iterations/level %d: %8.1f us  (%.2f ns per feature)" % (cnt, t, 1e3 * t / n))
c = np.array([r[0] for r in rows], float)
t = np.array([r[1] for r in rows], float)
slope, icpt = np.polyfit(c, t, 1)
print("n=%d features, %d levels, forward+backward: fixed part %.1f us (%.0f %% of a 2.7-iteration launch), %.1f us per iteration"
      % (n, maxlevel + 1, icpt, 100 * icpt / (icpt + 2.7 * slope), slope))
ctx.close()
