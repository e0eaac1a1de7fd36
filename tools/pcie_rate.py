#!/usr/bin/env python3
"""Raw rate of icelk_upload_gray_async (pinned host memory -> slot, copy stream) with nothing else on the device:
the ceiling of the PCIe-inclusive C3 figure on this box."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iceberg_tracking_code_amd import Context  # noqa: E402

w, h, n = 4000, 3000, 64
ctx = Context(w, h, n_slots=8, max_pts=1024)
ptrs = [ctx.host_alloc(w * h) for _ in range(4)]
for rep in range(3):
    ctx.sync()
    t0 = time.perf_counter()
    for i in range(n):
        ctx.upload_gray_async(i % 8, ptrs[i % 4], w, h, w)
    ctx.sync()
    dt = time.perf_counter() - t0
    print("%d uploads of %.1f MB: %.1f us each, %.1f GB/s, %.0f frames/s" % (n, w * h / 1e6, 1e6 * dt / n, n * w * h / dt / 1e9, n / dt))
for p in ptrs:
    ctx.host_free(p)
ctx.close()
