"""CPU oracle for the sparse Lucas-Kanade hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package,
and only as the checker or the timed CPU baseline.  The product
(iceberg_tracking_code_amd) never imports it.

PARITY UNPINNED: the reference's arithmetic for this path lives in OpenCV, which is neither in
/root/reference nor installed here, and the reference ships no fixtures for it.  See the header
of icelk_oracle.c.  The projection epilogue (utm_oracle.c) IS pinned: its arithmetic is the reference's own
numpy code and tests/golden/utm_golden.npz was produced by running it.
"""
from .cpu import (  # noqa: F401
    CRIT_COUNT, CRIT_EPS, FLAG_INITIAL_FLOW, FLAG_MIN_EIGENVALS,
    build, lib, set_threads, set_fb_distance, fb_distance, lk_stats, bgr2gray, pyrdown, pyramid_levels, build_pyramid, scharr,
    pyrlk, track_fb, min_eig_map, good_features, project_tracks, polygon_mask, points_in_polygon, grid_bin,
    VARIANTS, set_variant, get_variant, variants,
)
