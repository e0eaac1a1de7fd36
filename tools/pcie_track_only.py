#!/usr/bin/env python3
"""PCIe-fed loop with a long segment (track_len 16: one detection per 16 frames): how fast upload -> pyramid -> tracker
goes when the detector is (almost) out of the picture.  argv: depth [track_len]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iceberg_tracking_code_amd import Context, SegmentTracker, synth  # noqa: E402

w, h, K = 4000, 3000, 320
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 4
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ctx = Context(w, h, n_slots=depth + 6, max_pts=1 << 14)
fp = dict(maxCorners=10000, qualityLevel=0.007, minDistance=10, blockSize=10)
lk = dict(winSize=(21, 21), maxLevel=3, criteria=(3, 30, 0.01))
trk = SegmentTracker(w, h, T, feature_params=fp, lk_params=lk, ctx=ctx)
sh = synth.shifts(8, seed=1234)
pinned = []
for i in range(8):
    img = synth.frame(w, h, int(sh[i, 0]), int(sh[i, 1]), 1234)
    ptr = ctx.host_alloc(w * h)
    ctypes.memmove(ptr, img.ctypes.data, w * h)
    pinned.append(ptr)
o = [i % 8 if (i // 8) % 2 == 0 else 7 - i % 8 for i in range(K + depth + 1)]
for i in range(depth):
    trk.prefetch_pinned(pinned[o[i]], w)
ctx.sync()
t0 = time.perf_counter()
for k in range(K):
    trk.prefetch_pinned(pinned[o[k + depth]], w)
    trk.push_prefetched(wait=False)
ctx.sync()
el = time.perf_counter() - t0
print("depth %d track_len %d: %.1f pairs/s (%.1f us per frame)" % (depth, T, K / el, 1e6 * el / K))
trk.close()
