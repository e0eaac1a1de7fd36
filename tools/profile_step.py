#!/usr/bin/env python3
"""One C2-shaped tracking sequence for rocprofv3 (kernel trace or --pmc passes): a few frames, detection every
second frame, fused forward+backward LK.  Usage: rocprofv3 ... -- python3 tools/profile_step.py [c2|c5|ref] [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import CONFIGS, DETECT, TRACK_LEN, ping_pong  # noqa: E402
from iceberg_tracking_code_amd import Context, SegmentTracker, synth  # noqa: E402

cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
w, h = cfg["w"], cfg["h"]
ring = 4
ctx = Context(w, h, n_slots=ring, max_pts=max(cfg["max_corners"], 1 << 14) if cfg["max_corners"] > 0 else 1 << 18)
sh = synth.shifts(ring, seed=1234)
for i in range(ring):
    ctx.synth_frame(i, w, h, int(sh[i, 0]), int(sh[i, 1]), 1234)
trk = SegmentTracker(w, h, TRACK_LEN, dict(maxCorners=cfg["max_corners"], **DETECT),
                     dict(winSize=cfg["win"], maxLevel=cfg["max_level"], criteria=cfg["criteria"]), ctx=ctx)
for i in ping_pong(ring, steps):
    trk.push_slot(i, wait=False)
ctx.sync()
print("detect stats", ctx.detect_stats(), "live", trk.live())
ctx.close()
