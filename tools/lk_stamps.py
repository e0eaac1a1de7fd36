#!/usr/bin/env python3
"""Reads the file ICELK_LK_STAMPS=<file> leaves behind (entry / exit s_memtime and HW_ID of every workgroup of the last
segment tracker launch) and prints how the launch filled the chip: span, workgroup duration distribution, workgroups
resident over time, per-CU load."""
import sys

import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 3)
a = a[a[:, 0] != 0]
t0, t1, hw = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64), a[:, 2]
hw32 = (hw & np.uint64(0xffffffff)).astype(np.int64)
xcc = (hw >> np.uint64(32)).astype(np.int64)
for x in np.unique(xcc):   # every XCD counts from its own origin: align them at their first workgroup
    m = xcc == x
    base = t0[m].min()
    t0[m] -= base
    t1[m] -= base
dur = t1 - t0
span = t1.max()
wave_id, simd, cu, sh, se = hw32 & 15, (hw32 >> 4) & 3, (hw32 >> 8) & 15, (hw32 >> 12) & 1, (hw32 >> 13) & 7
print("workgroups %d, launch span %d cycles (%.1f us at 2.4 GHz)" % (len(a), span, span / 2400.0))
print("workgroup duration cycles: min %d  p10 %d  median %d  p90 %d  max %d" % (dur.min(), np.percentile(dur, 10),
                                                                              np.median(dur), np.percentile(dur, 90), dur.max()))
print("last entry at %d (%.0f%% of the span)" % (t0.max(), 100.0 * t0.max() / span))
edges = np.linspace(0, span, 21)
print("resident workgroups at 5% steps of the span:")
print("  " + " ".join("%d" % int(((t0 <= e) & (t1 > e)).sum()) for e in edges[:-1]))
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
uniq, cnt = np.unique(key, return_counts=True)
print("distinct (xcc, se, sh, cu): %d; workgroups per CU min %d median %d max %d" % (len(uniq), cnt.min(), int(np.median(cnt)), cnt.max()))
busy = np.zeros(len(uniq))
for i, k in enumerate(uniq):
    busy[i] = dur[key == k].sum()
print("sum of workgroup durations per CU / span: min %.2f median %.2f max %.2f (= mean workgroups resident per CU)" % (
    busy.min() / span, np.median(busy) / span, busy.max() / span))
print("first 16 entries (cycles):", np.sort(t0)[:16].tolist())
