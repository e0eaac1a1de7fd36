#!/usr/bin/env python3
"""One segment of 10 000 features tracked across the same frame pair eight times, launches back to back and alone on the
device: the single-pair tracker launch by itself (ICELK_NO_ORDER=1, ICELK_LK_STAMPS=<file> apply)."""
import sys, os
sys.path.insert(0, "/root/repo")
from iceberg_tracking_code_amd import Context, synth
w, h = 4000, 3000
ctx = Context(w, h, n_slots=3, max_pts=1 << 16)
sh = synth.shifts(3, seed=1234)
for i in range(3):
    ctx.synth_frame(i, w, h, int(sh[i, 0]), int(sh[i, 1]), 1234)
n = ctx.seg_detect(0, 10000, 0.007, 10, False, 10)
lk = dict(winSize=(21, 21), maxLevel=3, criteria=(3, 30, 0.01))
ctx.seg_track(0, 1, **lk)
ctx.prof_reset(); ctx.prof_enable(True)
for r in range(8):
    ctx.seg_track(r % 2, (r + 1) % 2, wait=False, **lk)
ctx.sync(); ctx.prof_enable(False)
print(os.environ.get("ICELK_NO_ORDER"), n, "features: %.1f us per launch" % ctx.prof_table()["lk_fb"]["avg_us"])
ctx.close()
