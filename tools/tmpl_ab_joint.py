#!/usr/bin/env python3
"""The JOINT tracker launch (last pair of a segment + first pair of the next) alone on the device, with and without the
template reuse (ICELK_NO_TEMPLATE_REUSE=1), timed with HIP events; and the same two pairs as two launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iceberg_tracking_code_amd import Context, synth
w, h = 4000, 3000
ctx = Context(w, h, n_slots=8, max_pts=1 << 14)
sh = synth.shifts(8, seed=1234)
for i in range(8):
    ctx.synth_frame(i, w, h, int(sh[i, 0]), int(sh[i, 1]), 1234)
lk = dict(winSize=(21, 21), maxLevel=3, criteria=(3, 30, 0.01))
det = (10000, 0.007, 10, False, 10)
ctx.seg_track_len_hint(2)
ctx.seg_detect(0, *det)
ctx.seg_track(0, 1, wait=False, **lk)
joint, two = [], []
for d in (2, 4):
    # detection of frame d staged while the segment of frame d-2 has its last pair (d-1, d) left
    ctx.seg_detect_begin(d, *det)
    ctx.seg_detect_stage(det[0])
    ctx.sync()
    ctx.prof_reset(); ctx.prof_enable(True)
    if d == 2:
        ctx.seg_track_defer(d - 1, d, **lk)
        ctx.seg_switch()
        ctx.seg_track(d, d + 1, wait=False, **lk)
    else:
        ctx.seg_track(d - 1, d, wait=False, **lk)
        ctx.seg_switch()
        ctx.seg_track(d, d + 1, wait=False, **lk)
    ctx.sync(); ctx.prof_enable(False)
    t = ctx.prof_table()
    if d == 2:
        joint.append(t["lk_fb_pair"]["avg_us"])
    else:
        two.append(1e3 * t["lk_fb"]["total_ms"])
print("reuse %s: joint launch %.1f us; the same two pairs as two launches %.1f us" % (
    "off" if os.environ.get("ICELK_NO_TEMPLATE_REUSE") else "on", joint[0], two[0]))
ctx.close()
