#!/usr/bin/env python3
"""Stamps of ONE joint tracker launch (ICELK_LK_STAMPS=<file> ICELK_LK_STAMPS_PAIR=1; the launch is made here): lifetime of
the workgroups of either job, and how many waves were resident over the launch."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iceberg_tracking_code_amd import Context, synth
w, h = 4000, 3000
path = os.environ["ICELK_LK_STAMPS"]
ctx = Context(w, h, n_slots=8, max_pts=1 << 15)
sh = synth.shifts(8, seed=1234)
for i in range(8):
    ctx.synth_frame(i, w, h, int(sh[i, 0]), int(sh[i, 1]), 1234)
lk = dict(winSize=(21, 21), maxLevel=3, criteria=(3, 30, 0.01))
det = (10000, 0.007, 10, False, 10)
ctx.seg_track_len_hint(2)
n0 = ctx.seg_detect(0, *det)
ctx.seg_track(0, 1, wait=False, **lk)
ctx.seg_detect_begin(2, *det)
n1 = ctx.seg_detect_stage(det[0])
ctx.sync()
ctx.seg_track_defer(1, 2, **lk)
ctx.seg_switch()
ctx.seg_track(2, 3, wait=False, **lk)
ctx.sync()
ctx.close()
a = np.fromfile(path, dtype=np.uint64).reshape(-1, 3)
b = np.arange(len(a))
ok = a[:, 0] != 0
G = b >> 3
g0, g1 = (n0 + 15) & ~7, (n1 + 15) & ~7
m = min(g0, g1) >> 3
mb = m & ~31   # the kernel's dealing: blocks of 32 groups alternate, the rest group by group (k_lk_fast.hip)
job = np.where(G < 2 * mb, (G >> 5) & 1, np.where(G < 2 * m, (G - 2 * mb) & 1, 1 if g0 < g1 else 0))
xcc = (a[:, 2] >> np.uint64(32)).astype(np.int64)
t0, t1 = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64)
hwid = (a[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
cukey = (xcc << 20) | (hwid & 0xff00)
for x in np.unique(cukey[ok]):   # s_memtime is consistent within a CU only: every CU counts from its own first entry
    mm = ok & (cukey == x)
    base = t0[mm].min()
    t0[mm] -= base; t1[mm] -= base
d = t1 - t0
span = t1[ok].max()
print("reuse %s: %d workgroups stamped, span %d ticks" % ("off" if os.environ.get("ICELK_NO_TEMPLATE_REUSE") else "on", int(ok.sum()), span))
for j in (0, 1):
    mm = ok & (job == j)
    print("  job %d (%s): %5d wgs, lifetime ticks median %d p90 %d max %d; sum/span = %.0f resident on average" % (
        j, "last pair of the old segment" if j == 0 else "first pair of the new", mm.sum(), np.median(d[mm]), np.percentile(d[mm], 90), d[mm].max(), d[mm].sum() / span))
edges = np.linspace(0, span, 21)[:-1]
print("  resident waves at 5% steps:", " ".join("%d" % int((ok & (t0 <= e) & (t1 > e)).sum()) for e in edges))
print("  last entry at %.0f%% of the span" % (100.0 * t0[ok].max() / span))
hw = (a[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
key = (xcc << 20) | (hw & 0xff00)
share = np.array([job[ok & (key == k)].mean() for k in np.unique(key[ok])])
print("  share of job-1 workgroups per CU: min %.2f median %.2f max %.2f over %d CUs" % (share.min(), np.median(share), share.max(), len(share)))
