#!/usr/bin/env python3
"""Host time of every Context call the SegmentTracker makes, over steady-state steps of the C2-shaped loop: mean duration
per call name, the share of wall time the host spends inside the library, and the call timeline of four steps."""
import collections
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iceberg_tracking_code_amd import Context, SegmentTracker, synth  # noqa: E402

w, h, ring, K = 4000, 3000, 24, 400
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
for _ in range(K + 12):
    o.append(i)
    if i + d < 0 or i + d >= ring:
        d = -d
    i += d
marks = []
pcie = len(sys.argv) > 1 and sys.argv[1] == "pcie"     # frames from pinned host memory, `depth` uploads ahead
if pcie:
    import ctypes
    depth = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    trk.close()
    ctx = Context(w, h, n_slots=depth + 6, max_pts=1 << 14)
    trk = SegmentTracker(w, h, 2, feature_params=fp, lk_params=lk, ctx=Logged(ctx))
    from iceberg_tracking_code_amd import synth as _s
    pinned = []
    for i in range(8):
        img = _s.frame(w, h, int(sh[i, 0]), int(sh[i, 1]), 1234)
        ptr = ctx.host_alloc(w * h)
        ctypes.memmove(ptr, img.ctypes.data, w * h)
        pinned.append(ptr)
    o = [i % 8 if (i // 8) % 2 == 0 else 7 - i % 8 for i in range(K + depth + 1)]
    for i in range(depth):
        trk.prefetch_pinned(pinned[o[i]], w)
t0 = time.perf_counter()
for k in range(K):
    marks.append((k, len(log), time.perf_counter()))
    if pcie:
        trk.prefetch_pinned(pinned[o[k + depth]], w)
        trk.push_prefetched(wait=False)
    else:
        trk.push_slot(o[k], False, *o[k + 1:k + 7])
ctx.sync()
el = time.perf_counter() - t0
first = marks[40][1]
per = collections.defaultdict(list)
for name, a, b in log[first:]:
    per[name].append(b - a)
print("%.1f pairs/s; %.1f us per step" % (K / el, 1e6 * el / K))
tot = 0.0
for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    tot += sum(v)
    print("  %-24s n=%4d mean %7.1f us  total/step %6.1f us" % (name, len(v), 1e6 * sum(v) / len(v), 1e6 * sum(v) / (K - 40)))
print("host inside the library: %.1f us per step (%.0f%% of wall)" % (1e6 * tot / (K - 40), 100 * tot / (el * (K - 40) / K)))
k0 = 200
base = marks[k0][2]
for k in range(k0, k0 + 4):
    print("step %d at %.1f us" % (k, 1e6 * (marks[k][2] - base)))
    for name, a, b in log[marks[k][1]:marks[k + 1][1]]:
        print("    %8.1f .. %8.1f  %s" % (1e6 * (a - base), 1e6 * (b - base), name))
trk.close()
