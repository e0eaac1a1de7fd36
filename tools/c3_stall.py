#!/usr/bin/env python3
"""Which library call holds the host when the PCIe-fed loop of C3 loses ~9 ms in one step (some runs do, some do not:
profiles/r04_c3_steps.txt).  The loop of bench.py's PCIe-inclusive leg -- a fresh handle, 10 warm-up frames, 65 frames
from pinned memory with 6 uploads ahead -- several times in one process, every Context call timed."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from iceberg_tracking_code_amd import Context, SegmentTracker, synth  # noqa: E402

w, h, NF, W, depth, spare = 4000, 3000, 65, 10, 6, 6
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
fp = dict(maxCorners=10000, qualityLevel=0.007, minDistance=10, blockSize=10)
lk = dict(winSize=(21, 21), maxLevel=3, criteria=(3, 30, 0.01))
sh = synth.shifts(24, seed=1234)
c0 = Context(w, h, n_slots=2, max_pts=1 << 14)
pinned = []
for i in range(24):
    img = synth.frame(w, h, int(sh[i, 0]), int(sh[i, 1]), 1234)
    p = c0.host_alloc(w * h)
    ctypes.memmove(p, img.ctypes.data, w * h)
    pinned.append(p)
c0.close()
order = [i % 24 if (i // 24) % 2 == 0 else 23 - i % 24 for i in range(NF + W)]


class Logged:
    def __init__(self, inner, log):
        self._inner, self._log = inner, log

    def __getattr__(self, name):
        f = getattr(self._inner, name)
        if not callable(f):
            return f

        def g(*a, **k):
            t0 = time.perf_counter()
            r = f(*a, **k)
            self._log.append((name, t0, time.perf_counter()))
            return r
        return g


for rep in range(reps):
    log = []
    ctx = Context(w, h, n_slots=depth + spare, max_pts=1 << 14)
    lc = Logged(ctx, log)
    hw = SegmentTracker(w, h, 2, feature_params=fp, lk_params=lk, ctx=lc, lookahead=False)
    for i in range(W):
        hw.push_pinned(pinned[order[i]], w, wait=False)
    hw.abort()
    ht = SegmentTracker(w, h, 2, feature_params=fp, lk_params=lk, ctx=lc)
    ctx.sync()
    del log[:]
    fr = [pinned[order[W + i]] for i in range(NF)]
    t0 = time.perf_counter()
    for i in range(depth):
        ht.prefetch_pinned(fr[i], w)
    marks = []
    for i in range(NF):
        marks.append(len(log))
        if i + depth < NF:
            ht.prefetch_pinned(fr[i + depth], w)
        ht.push_prefetched(wait=False)
    ctx.sync()
    el = time.perf_counter() - t0
    worst = sorted(log, key=lambda e: e[1] - e[2])[:3]
    step_of = lambda t: max(k for k in range(NF) if k == 0 or log[marks[k]][1] <= t) if marks else -1
    print("rep %d: %.0f pairs/s (%.1f ms); longest calls: %s" % (rep, (NF - 1) / el, 1e3 * el, "; ".join(
        "%s %.0f us at step %d (+%.1f ms)" % (n, 1e6 * (b - a), step_of(a), 1e3 * (a - t0)) for n, a, b in worst)))
    ctx.close()
