#!/usr/bin/env python3
"""The driver's short run (5 warm-up steps, a device-wide wait, 20 timed steps, a wait) seen from the host: when every
Context call of the SegmentTracker starts and how long it takes, step by step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iceberg_tracking_code_amd import Context, SegmentTracker, synth  # noqa: E402
w, h, ring, W, K = 4000, 3000, 64, 5, 20
ctx = Context(w, h, n_slots=ring, max_pts=1 << 14)
sh = synth.shifts(ring, seed=1234)
for i in range(ring):
    ctx.synth_frame(i, w, h, int(sh[i, 0]), int(sh[i, 1]), 1234)
ctx.sync()
log = []


class Logged:
    def __init__(self, inner):
        self._inner = inner

    def __getattr__(self, name):
        f = getattr(self._inner, name)
        if not callable(f):
            return f

        def g(*a, **k):
            t0 = time.perf_counter()
            r = f(*a, **k)
            log.append((name, t0, time.perf_counter()))
            return r
        return g


fp = dict(maxCorners=10000, qualityLevel=0.007, minDistance=10, blockSize=10)
lk = dict(winSize=(21, 21), maxLevel=3, criteria=(3, 30, 0.01))
trk = SegmentTracker(w, h, 2, feature_params=fp, lk_params=lk, ctx=Logged(ctx))
o, i, d = [], 0, 1
for _ in range(W + K + 12):
    o.append(i)
    if i + d < 0 or i + d >= ring:
        d = -d
    i += d
for k in range(W):
    trk.push_slot(o[k], False, *o[k + 1:k + 7])
ctx.sync()
n0 = len(log)
t0 = time.perf_counter()
marks = []
for k in range(W, W + K):
    marks.append((k, len(log), time.perf_counter()))
    trk.push_slot(o[k], False, *o[k + 1:k + 7])
ctx.sync()
el = time.perf_counter() - t0
print("%d timed steps in %.1f us: %.0f pairs/s" % (K, 1e6 * el, K / el))
marks.append((W + K, len(log), time.perf_counter()))
for j in range(K):
    k, a, t = marks[j]
    print("step %d at %.1f us" % (k, 1e6 * (t - t0)))
    for name, s, e in log[a:marks[j + 1][1]]:
        if e - s > 4e-6 or name in ("seg_track", "seg_track_defer"):
            print("    %8.1f .. %8.1f  %-22s %6.1f us" % (1e6 * (s - t0), 1e6 * (e - t0), name, 1e6 * (e - s)))
trk.abort()
trk.close()
