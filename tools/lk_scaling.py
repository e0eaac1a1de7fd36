#!/usr/bin/env python3
"""Kernel time of the fused forward+backward LK launch versus the number of features (occupancy/tail study)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iceberg_tracking_code_amd import Context, synth  # noqa: E402

w, h = 4000, 3000
win = (int(sys.argv[1]), int(sys.argv[1])) if len(sys.argv) > 1 else (21, 21)
maxlevel = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = Context(w, h, n_slots=2, max_pts=1 << 16)
sh = synth.shifts(3, seed=1234)
ctx.synth_frame(0, w, h, int(sh[1, 0]), int(sh[1, 1]), 1234)
ctx.synth_frame(1, w, h, int(sh[2, 0]), int(sh[2, 1]), 1234)
pts = ctx.good_features(0, 40000, 0.007, 10, False, 10).reshape(-1, 2)
print("corners", len(pts))
ctx.track_fb(0, 1, pts[:1000], win, maxlevel)
for n in (256, 1024, 2048, 3072, 4096, 6144, 8192, 9216, 10000, 12288, 16384, 20000, 30000, 40000):
    if n > len(pts):
        break
    ctx.prof_reset()
    ctx.prof_enable(True)
    for _ in range(5):
        ctx.track_fb(0, 1, pts[:n], win, maxlevel)
    ctx.prof_enable(False)
    t = ctx.prof_table()["lk_fb"]
    print("n=%6d  lk_fb avg %8.1f us   %.2f ns/feature" % (n, t["avg_us"], 1e3 * t["avg_us"] / n))
ctx.close()
