#!/usr/bin/env python3
"""Which features make the long workgroups of a tracker launch?  Run with ICELK_NO_ORDER=1 ICELK_LK_KERNEL=one
ICELK_LK_STAMPS=<file>: workgroup b then tracks corner b, and the durations can be set against the corner's position."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iceberg_tracking_code_amd import Context, synth

w, h = 4000, 3000
path = os.environ["ICELK_LK_STAMPS"]
ctx = Context(w, h, n_slots=4, max_pts=1 << 14)
sh = synth.shifts(4, seed=1234)
for i in range(3):
    ctx.synth_frame(i, w, h, int(sh[i, 0]), int(sh[i, 1]), 1234)
pts = ctx.good_features(0, 10000, 0.007, 10, False, 10).reshape(-1, 2)
n = ctx.seg_detect(0, 10000, 0.007, 10, False, 10)
lk = ((21, 21), 3, (3, 30, 0.01))
ctx.seg_track(0, 1, *lk)
ctx.seg_track(1, 2, *lk)   # warm
ctx.close()
a = np.fromfile(path, dtype=np.uint64).reshape(-1, 3)[:n]
d = (a[:, 1].astype(np.int64) - a[:, 0].astype(np.int64)) / 2400.0
ok = a[:, 0] != 0
print("n", n, "tracked", int(ok.sum()), "median %.1f us  p90 %.1f  p99 %.1f  max %.1f" % (np.median(d[ok]), np.percentile(d[ok], 90), np.percentile(d[ok], 99), d[ok].max()))
border = np.minimum(np.minimum(pts[:, 0], w - 1 - pts[:, 0]), np.minimum(pts[:, 1], h - 1 - pts[:, 1]))
for lo, hi in ((0, 20), (20, 40), (40, 80), (80, 160), (160, 1e9)):
    m = ok & (border >= lo) & (border < hi)
    if m.sum():
        print("distance to the frame border %4d..%-6d: %5d features, mean %.1f us, p90 %.1f us" % (lo, min(hi, 9999), m.sum(), d[m].mean(), np.percentile(d[m], 90)))
hw = (a[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
xcc = (a[:, 2] >> np.uint64(32)).astype(np.int64)
t0 = a[:, 0].astype(np.int64)
for x in range(8):
    m = ok & (xcc == x)
    if m.sum():
        tt = (t0[m] - t0[m].min()) / 2400.0
        late = tt > np.percentile(tt, 80)
        print("xcc %d: %d wgs, span of entries %.1f us, mean duration early 80%% %.1f us, last 20%% %.1f us" % (x, m.sum(), tt.max(), d[m][~late].mean(), d[m][late].mean()))
