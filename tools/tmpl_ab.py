#!/usr/bin/env python3
"""Template reuse, launch by launch: one segment of 10 000 features, pair 1 (leaves templates unless ICELK_NO_TEMPLATE_REUSE)
and pair 2 (takes them), each alone on the device, timed with HIP events; repeated over fresh segments."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iceberg_tracking_code_amd import Context, synth
w, h = 4000, 3000
ctx = Context(w, h, n_slots=4, max_pts=1 << 14)
sh = synth.shifts(4, seed=1234)
for i in range(3):
    ctx.synth_frame(i, w, h, int(sh[i, 0]), int(sh[i, 1]), 1234)
lk = dict(winSize=(21, 21), maxLevel=3, criteria=(3, 30, 0.01))
ctx.seg_track_len_hint(2)
t1, t2 = [], []
for rep in range(6):
    n = ctx.seg_detect(0, 10000, 0.007, 10, False, 10)
    ctx.sync()
    for pair, acc in ((0, t1), (1, t2)):
        ctx.prof_reset(); ctx.prof_enable(True)
        ctx.seg_track(pair, pair + 1, wait=False, **lk)
        ctx.sync(); ctx.prof_enable(False)
        acc.append(ctx.prof_table()["lk_fb"]["avg_us"])
live, _ = ctx.seg_live()
print("reuse %s: %d features, %d alive after two pairs; pair 1 %.1f us, pair 2 %.1f us (medians of 6)" % (
    "off" if os.environ.get("ICELK_NO_TEMPLATE_REUSE") else "on", n, live, sorted(t1)[3], sorted(t2)[3]))
ctx.close()
