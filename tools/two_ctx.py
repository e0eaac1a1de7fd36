#!/usr/bin/env python3
"""Experiment: do two independent tracker pipelines on one GPU fill each other's launch gaps and tails?
One context with 10 000 features versus two contexts with 5 000 features each, driven alternately from one thread."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iceberg_tracking_code_amd import Context, SegmentTracker, synth  # noqa: E402

w, h, ring, K, W = 4000, 3000, 12, 200, 10


def make(maxc, seed):
    ctx = Context(w, h, n_slots=ring, max_pts=1 << 14)
    sh = synth.shifts(ring, seed=seed)
    for i in range(ring):
        ctx.synth_frame(i, w, h, int(sh[i, 0]), int(sh[i, 1]), seed)
    ctx.sync()
    fp = dict(maxCorners=maxc, qualityLevel=0.007, minDistance=10, blockSize=10)
    lk = dict(winSize=(21, 21), maxLevel=3, criteria=(3, 30, 0.01))
    return SegmentTracker(w, h, 2, feature_params=fp, lk_params=lk, ctx=ctx)


def order(n):
    out, i, d = [], 0, 1
    for _ in range(n):
        out.append(i)
        if i + d < 0 or i + d >= ring:
            d = -d
        i += d
    return out


def run(trackers):
    o = order(K + W + 2)
    for i in range(W):
        for t in trackers:
            t.push_slot(o[i], wait=False, next_slot=o[i + 1], next2_slot=o[i + 2])
    for t in trackers:
        t.ctx.sync()
    f0 = [t.live()[1] for t in trackers]
    t0 = time.perf_counter()
    for i in range(W, W + K):
        for t in trackers:
            t.push_slot(o[i], wait=False, next_slot=o[i + 1], next2_slot=o[i + 2])
    for t in trackers:
        t.ctx.sync()
    dt = time.perf_counter() - t0
    feats = sum(t.live()[1] - a for t, a in zip(trackers, f0))
    return K / dt, feats / dt


one = [make(10000, 1234)]
print("1 x 10000 features: %.1f steps/s, %.2f M features/s" % tuple(v * s for v, s in zip(run(one), (1, 1e-6))))
one[0].close()
two = [make(5000, 1234), make(5000, 99)]
print("2 x  5000 features: %.1f steps/s each, %.2f M features/s" % tuple(v * s for v, s in zip(run(two), (1, 1e-6))))
for t in two:
    t.close()
