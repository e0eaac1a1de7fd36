#!/usr/bin/env python3
"""End-to-end rate of the folder driver (SURVEY.md 8f-3, s1_lucaskanade_tracking.py:272,310-311): `track_image_sequence` on a
folder of 12 MP JPEGs -- host decode (PIL, N threads decoding ahead), crop-on-upload, gray conversion, detection,
tracking, segment read-out -- in frames per second, beside the decode rate of one thread and the device loop's own rate.

    python tools/e2e_sequence.py [n_frames] > profiles/r03_e2e_sequence.txt
"""
import datetime as dt
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from PIL import Image  # noqa: E402
from iceberg_tracking_code_amd import Context, synth, track_image_sequence  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
w, h, T, dts = 4000, 3000, 2, 60
tmp = tempfile.mkdtemp(prefix="icelk_e2e_")
src, dst = os.path.join(tmp, "photos"), os.path.join(tmp, "tracks")
os.makedirs(src)
os.makedirs(dst)
ctx = Context(w, h, n_slots=1, max_pts=64)
sh = synth.shifts(n, seed=1234)
t0 = dt.datetime(2019, 7, 24, 10, 0, 0)
names = []
for k in range(n):
    ctx.synth_frame(0, w, h, int(sh[k, 0]), int(sh[k, 1]), 1234)
    g = ctx.download_level(0, 0)
    p = os.path.join(src, (t0 + dt.timedelta(seconds=k * dts)).strftime("%Y%m%d-%H%M%S") + ".jpg")
    Image.fromarray(np.stack([g, g, g], 2)).save(p, quality=92)
    names.append(p)
ctx.close()
size = sum(os.path.getsize(p) for p in names) / n / 1e6
t = time.perf_counter()
for p in names[:6]:
    np.array(Image.open(p))
dec = (time.perf_counter() - t) / 6
print("%d frames of %dx%d, %.1f MB per JPEG; PIL decode %.1f ms per frame on one thread (%.1f frames/s)" % (n, w, h, size, 1e3 * dec, 1 / dec))
fp = dict(maxCorners=10000, qualityLevel=0.007, minDistance=10, blockSize=10)
lk = dict(winSize=(21, 21), maxLevel=3, criteria=(3, 30, 0.01))
cores = len(os.sched_getaffinity(0))
for threads in (1, 2, 4, 8, 16):
    if threads > max(cores, 1) * 2:
        break
    best = None
    for rep in range(2):
        t = time.perf_counter()
        out = track_image_sequence(names, dst, T, dts, feature_params=fp, lk_params=lk, decode_threads=threads,
                                   decode_ahead=max(6, 2 * threads), save=False)
        el = time.perf_counter() - t
        best = el if best is None else min(best, el)
    print("decode_threads %2d: %6.1f frames/s end to end (%d segments of %d..%d tracks; %.0f ms per frame; host has %d usable cores)"
          % (threads, n / best, len(out), min(len(s[1]) for s in out), max(len(s[1]) for s in out), 1e3 * best / n, cores))
for p in names:
    os.remove(p)
