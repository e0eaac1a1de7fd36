#!/usr/bin/env python3
"""The pyramid kernel(s) alone on the device, 30 builds back to back (for rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iceberg_tracking_code_amd import Context
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4000, 3000)
ml = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctx = Context(w, h, n_slots=4, max_pts=1024)
for i in range(4):
    ctx.synth_frame(i, w, h, 100 * i, -50 * i, 1234)
ctx.sync()
ahead = os.environ.get("PYR_AHEAD") == "1"     # the copy-stream path (one-wave geometry unless ICELK_PYR_AHEAD_WIDE=1)
for rep in range(30):
    ctx.drop_pyramid(rep % 4)
    if ahead:
        ctx.build_pyramid_ahead(rep % 4, (21, 21), ml)
    else:
        ctx.build_pyramid(rep % 4, (21, 21), ml)
ctx.sync()
ctx.close()
