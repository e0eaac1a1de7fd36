#!/usr/bin/env python3
"""Iterations per feature on the bench data (C2): histogram, share of the launch's iterations in its slowest features,
and how well a track's cost in one pair predicts its cost in the next pair of the same segment."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iceberg_tracking_code_amd import Context, synth

w, h = 4000, 3000
ctx = Context(w, h, n_slots=4, max_pts=1 << 14)
sh = synth.shifts(4, seed=1234)
for i in range(3):
    ctx.synth_frame(i, w, h, int(sh[i, 0]), int(sh[i, 1]), 1234)
n = ctx.seg_detect(0, 10000, 0.007, 10, False, 10)
ctx.prof_enable(True)
lk = ((21, 21), 3, (3, 30, 0.01))
ctx.seg_track(0, 1, *lk)
f1, b1 = ctx.prof_iterations()
import ctypes
raw1 = np.zeros(n, np.uint32); k = ctypes.c_int(0)
ctx._lib.icelk_prof_iterations(ctx._h, raw1.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), n, ctypes.byref(k))
ctx.seg_track(1, 2, *lk)
raw2 = np.zeros(n, np.uint32)
ctx._lib.icelk_prof_iterations(ctx._h, raw2.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), n, ctypes.byref(k))
t1 = (raw1 & 0xffff).astype(int) + (raw1 >> 16).astype(int)
alive2 = raw2 != 0xffffffff
t2 = (raw2 & 0xffff).astype(int) + (raw2 >> 16).astype(int)
print("features", n, "pair 1: mean iterations fwd %.2f bwd %.2f (4 levels each)" % (f1.mean(), b1.mean()))
print("histogram of fwd+bwd iterations (pair 1), bins of 8:", np.bincount(np.minimum(t1 // 8, 31)).tolist())
s = np.sort(t1)[::-1]
for p in (1, 5, 10, 20):
    print("slowest %d%% of the features run %.1f%% of the iterations, threshold %d" % (p, 100.0 * s[:n * p // 100].sum() / s.sum(), s[n * p // 100]))
a, b = t1[alive2], t2[alive2]
print("tracks alive in pair 2:", int(alive2.sum()), "corr(pair1, pair2) = %.3f" % np.corrcoef(a, b)[0, 1])
thr = np.percentile(a, 90)
slow2 = b >= np.percentile(b, 90)
print("of the slowest 10%% in pair 2, %.0f%% were among the slowest 10%% in pair 1; of the slowest 5%%: %.0f%% in pair-1 top 10%%" % (
    100.0 * (a[slow2] >= thr).mean(), 100.0 * (a[b >= np.percentile(b, 95)] >= thr).mean()))
ctx.close()
