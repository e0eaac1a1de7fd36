#!/usr/bin/env python3
"""Throughput of the projection epilogue (k_project_tracks): tracks/s on the GPU (kernel alone via HIP events, and
through the C ABI with its PCIe transfers) beside the CPU oracle on the host cores (1 thread, scalar C).
Usage: python tests/utm_bench.py [n_tracks] [n_vertices]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle  # noqa: E402
import utm_golden as G  # noqa: E402
from iceberg_tracking_code_amd import Context, project_tracks  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
nv = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rng = np.random.default_rng(5)
t = np.zeros((n, nv, 2))
t[:, 0] = np.stack([rng.uniform(0, 3456, n), rng.uniform(150, 1300, n)], 1)
t[:, 1:] = t[:, :1] + np.cumsum(rng.normal(0, 0.3, (n, nv - 1, 2)), 1)
t = t.astype(np.float32)
z = G.load()
cam = G.camera(z, 0.3)
f = dict(max_speed=1.7, min_speed=0.0, max_speedfactor=2.5, max_angle=60, speed_threshold=0.1)
ctx = Context(64, 64, n_slots=1, max_pts=max(n, 1 << 14))
project_tracks(ctx, t, cam, 60, **f)
ctx.prof_reset()
ctx.prof_enable(True)
t0 = time.perf_counter()
reps = 20
for _ in range(reps):
    r = project_tracks(ctx, t, cam, 60, **f)
t1 = time.perf_counter()
ctx.prof_enable(False)
k = ctx.prof_table()["project_tracks"]
c0 = time.perf_counter()
w = oracle.project_tracks(t, cam.as_dict(), dict(interval_s=60, **f))
c1 = time.perf_counter()
alg = n * (8 * nv + 40 * (nv - 1) + 1)
print(json.dumps({
    "tracks": n, "vertices": nv, "kernel_us": k["avg_us"], "kernel_tracks_per_s": n / (k["avg_us"] * 1e-6),
    "algorithmic_bytes": alg, "kernel_GBps": alg / (k["avg_us"] * 1e-6) / 1e9,
    "abi_tracks_per_s_with_pcie": n * reps / (t1 - t0), "oracle_1thread_tracks_per_s": n / (c1 - c0),
    "kept_fraction": float(r["keep"].mean()), "equal_to_oracle": bool(np.array_equal(r["speed"], w["speed"]))}))
ctx.close()
