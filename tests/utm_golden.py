"""Shared loader for tests/golden/utm_golden.npz (outputs of the reference's own s2_cam_to_utm / camtools code,
produced by tests/golden/make_utm_golden.py).  Loaded with allow_pickle=False."""
import os

import numpy as np

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "utm_golden.npz")
SCENARIOS = ("s0", "s1", "s2")
INT_KEYS = ("image_width", "image_height", "crop_left", "crop_right", "crop_top", "crop_bottom")


def load():
    return np.load(PATH, allow_pickle=False)


def calib(z):
    d = {str(k): float(v) for k, v in zip(z["calib_keys"], z["calib_values"])}
    for k in INT_KEYS:
        d[k] = int(d[k])
    return d


def camera(z, tide):
    from iceberg_tracking_code_amd.utm import CameraModel
    return CameraModel(tide_elevation=float(tide), **calib(z))


def scenario(z, name):
    """inputs [(file name, tracks, tide)], outputs [(file name, {x,y,u,v,speed,time})], filter dict."""
    ins = [(str(z["%s_in_%02d_name" % (name, k)]), z["%s_in_%02d_tracks" % (name, k)],
            float(z["%s_in_%02d_tide" % (name, k)])) for k in range(int(z["%s_n_in" % name]))]
    outs = [(str(z["%s_out_%02d_name" % (name, k)]),
             {key: z["%s_out_%02d_%s" % (name, k, key)] for key in ("x", "y", "u", "v", "speed", "time")})
            for k in range(int(z["%s_n_out" % name]))]
    f = z["%s_filter" % name]
    filt = dict(max_speed=f[0], min_speed=f[1], max_speedfactor=f[2], max_angle=f[3], speed_threshold=f[4])
    return ins, outs, filt


def same_bits(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()
