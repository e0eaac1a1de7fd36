#!/usr/bin/env python3
"""Steady-state steps of the C2-shaped loop with every Context call, its arguments' first entries and its result: which
detection is begun / prepared / adopted at which step, and when (host clock)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iceberg_tracking_code_amd import Context, SegmentTracker, synth  # noqa: E402
w, h, ring, K = 4000, 3000, 24, 120
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
            log.append((name, t0, time.perf_counter(), a[:2], r if isinstance(r, (int, type(None))) else "."))
            return r
        return g


fp = dict(maxCorners=10000, qualityLevel=0.007, minDistance=10, blockSize=10)
lk = dict(winSize=(21, 21), maxLevel=3, criteria=(3, 30, 0.01))
trk = SegmentTracker(w, h, 2, feature_params=fp, lk_params=lk, ctx=Logged(ctx))
o, i, d = [], 0, 1
for _ in range(K + 12):
    o.append(i)
    if i + d < 0 or i + d >= ring:
        d = -d
    i += d
marks = []
for k in range(K):
    marks.append((k, len(log), time.perf_counter()))
    trk.push_slot(o[k], False, *o[k + 1:k + 7])
ctx.sync()
k0 = 80
base = marks[k0][2]
for k in range(k0, k0 + 6):
    print("step %d (slot %d) at %.1f us   queue %s" % (k, o[k], 1e6 * (marks[k][2] - base), ""))
    for name, a, b, args, r in log[marks[k][1]:marks[k + 1][1]]:
        print("    %8.1f .. %8.1f  %-22s %s -> %s" % (1e6 * (a - base), 1e6 * (b - base), name, args, r))
print(ctx.seg_tail_stats())
trk.close()
