#!/usr/bin/env python3
"""Does a kernel that runs beside MANY tiny kernels of another stream get slower -- beyond what they compute?  Every kernel
boundary carries cache maintenance (the L2s of the eight XCDs are not coherent with each other), and a period of the pipeline
has ~17 of them.  A tracker launch (10 000 features, forward + backward, 4000x3000) alone, then with 400 one-element fills
of a side stream enqueued right before it; HIP-event duration of the tracker launch, and of the pyramid kernel likewise."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from iceberg_tracking_code_amd import Context, synth

w, h = 4000, 3000
ctx = Context(w, h, n_slots=2, max_pts=1 << 14)
ctx.synth_frame(0, w, h, 0, 0, 1234)
ctx.synth_frame(1, w, h, 300, -200, 1234)
ctx.build_pyramid(0, (21, 21), 3)
ctx.build_pyramid(1, (21, 21), 3)
ctx.sync()
rng = np.random.RandomState(1)
pts = np.stack([rng.uniform(30, w - 30, 10000), rng.uniform(30, h - 30, 10000)], 1).astype(np.float32)
side = torch.cuda.Stream()
x = torch.zeros(64, device="cuda")
for n_side in (0, 400, 0, 400, 0, 1500):
    ctx.prof_reset()
    ctx.prof_enable(True)
    for rep in range(12):
        with torch.cuda.stream(side):
            for _ in range(n_side):
                x.add_(1.0)
        ctx.track_fb(0, 1, pts, (21, 21), 3, (3, 30, 0.01))
        side.synchronize()
    ctx.sync()
    ctx.prof_enable(False)
    t = ctx.prof_table()
    lk = [v for k, v in t.items() if k.startswith("lk")]
    print("tiny kernels beside it: %4d   tracker launch %.1f us (%d launches)" % (n_side, lk[0]["avg_us"], lk[0]["launches"]))
for n_side in (0, 400, 0, 400):
    ctx.prof_reset()
    ctx.prof_enable(True)
    for rep in range(12):
        with torch.cuda.stream(side):
            for _ in range(n_side):
                x.add_(1.0)
        ctx.drop_pyramid(rep % 2)
        ctx.build_pyramid(rep % 2, (21, 21), 3)
        ctx.sync()
        side.synchronize()
    ctx.prof_enable(False)
    t = ctx.prof_table()
    print("tiny kernels beside it: %4d   pyramid %s" % (n_side, {k: round(v["avg_us"], 1) for k, v in t.items() if "pyr" in k}))
ctx.close()
