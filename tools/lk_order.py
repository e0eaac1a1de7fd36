#!/usr/bin/env python3
"""Kernel time of the fused LK launch versus the ORDER in which features are handed to workgroups.

strength : detector order (what the reference loop produces)
morton   : Z-order of the level-0 position
xcd      : Z-order, then dealt so that the 8 workgroups dispatched round-robin to the 8 XCDs each walk their own
           contiguous eighth of the Z-curve (workgroup b -> sorted[(b % 8) * n/8 + b / 8])
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iceberg_tracking_code_amd import Context, synth  # noqa: E402


def part1by1(v):
    v = v.astype(np.uint64) & 0xFFFF
    v = (v | (v << 8)) & 0x00FF00FF
    v = (v | (v << 4)) & 0x0F0F0F0F
    v = (v | (v << 2)) & 0x33333333
    v = (v | (v << 1)) & 0x55555555
    return v


def morton(pts, shift=0):
    x = pts[:, 0].astype(np.int64) >> shift
    y = pts[:, 1].astype(np.int64) >> shift
    return np.argsort(part1by1(x) | (part1by1(y) << 1), kind="stable")


def xcd_deal(order, nx=8):
    n = len(order)
    chunk = (n + nx - 1) // nx
    out = np.empty(n, order.dtype)
    b = np.arange(n)
    src = (b % nx) * chunk + b // nx
    ok = src < n
    # ragged tail: fall back to plain order for the few blocks whose source is past the end
    out[ok] = order[src[ok]]
    used = np.zeros(n, bool)
    used[src[ok]] = True
    out[~ok] = order[~used]
    return out


w, h = 4000, 3000
win = (int(sys.argv[1]), int(sys.argv[1])) if len(sys.argv) > 1 else (21, 21)
maxlevel = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
only = sys.argv[4] if len(sys.argv) > 4 else None
ctx = Context(w, h, n_slots=2, max_pts=1 << 16)
sh = synth.shifts(3, seed=1234)
ctx.synth_frame(0, w, h, int(sh[1, 0]), int(sh[1, 1]), 1234)
ctx.synth_frame(1, w, h, int(sh[2, 0]), int(sh[2, 1]), 1234)
pts = ctx.good_features(0, n, 0.007, 10, False, 10).reshape(-1, 2)
print("corners", len(pts))
orders = {"strength": np.arange(len(pts))}
orders["morton"] = morton(pts)
orders["xcd"] = xcd_deal(orders["morton"])
orders["xcd_rowband"] = xcd_deal(np.lexsort((pts[:, 0], pts[:, 1].astype(np.int64) // 64)))
ref = None
for name, o in orders.items():
    if only and name != only:
        continue
    p = np.ascontiguousarray(pts[o])
    ctx.track_fb(0, 1, p, win, maxlevel)
    ctx.prof_reset()
    ctx.prof_enable(True)
    for _ in range(10):
        res = ctx.track_fb(0, 1, p, win, maxlevel)
    ctx.prof_enable(False)
    t = ctx.prof_table()["lk_fb"]
    inv = np.empty_like(o)
    inv[o] = np.arange(len(o))
    p1 = res["p1"][inv]
    if ref is None:
        ref = p1
    print("%-12s lk_fb avg %8.1f us  same=%s" % (name, t["avg_us"], bool(np.array_equal(ref, p1))))
ctx.close()
