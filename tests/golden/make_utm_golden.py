#!/usr/bin/env python3
"""Generates tests/golden/utm_golden.npz by RUNNING THE REFERENCE (development container only).

What runs: the reference's own `s2_cam_to_utm.cam_to_utm()` (s2_cam_to_utm.py:163-363) and
`imports.camtools.Camera.photo_to_utm` / `photocords_cropped_to_uncropped` (camtools.py:286-332, 414-421), imported
from /root/reference, on synthetic track files written here.  What is committed: only this script and the data it
produced (inputs + the reference's outputs) -- no reference source.

Two things the reference needs are absent from the image and are NOT emulated:
  * `shapefile` (pyshp): camtools.py:16 imports it at module level; none of the functions used here touches it, so an
    empty placeholder module is registered to let the import statement pass;
  * the calibration workbook (openpyxl) and the pickled tide series that `Camera.__init__` reads
    (camtools.py:111-179; pickles from the reference are never loaded): `ct.Camera` is replaced by a factory that
    returns a real `Camera` instance created with `__new__` whose `cam` / `pic` dicts are filled with the synthetic
    calibration below, using the same expressions and numpy scalar types as `__init__` (np.radians of the angles,
    sigma = width / chipsize * sigma, H = elevation - antenna_height - float(tide)).

Usage (development container):  python tests/golden/make_utm_golden.py
"""
import datetime as dt
import glob
import os
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "utm_golden.npz")

CALIB = dict(image_width=3456, image_height=2304, sensor_width=22.3, easting=497812.37, northing=6521034.81,
             elevation=431.62, antenna_height=1.35, theta=201.4, phi=-11.85, psi=1.27, sigma=24.6,
             crop_left=0, crop_right=0, crop_top=1000, crop_bottom=4)


def tide_of(stamp):
    """Synthetic tide elevation (m) for a '%Y%m%d-%H%M%S' stamp, one value per minute like the reference's series."""
    t = dt.datetime.strptime(stamp, "%Y%m%d-%H%M%S")
    minute = t.hour * 60 + t.minute
    return float(np.float32(1.8 * np.sin(2 * np.pi * minute / 745.0) + 0.3))


def camera_dicts(stamp):
    """cam / pic exactly as Camera.__init__ builds them (camtools.py:127-179), from CALIB and tide_of(stamp)."""
    p = {k: (np.int64(v) if isinstance(v, int) else np.float64(v)) for k, v in CALIB.items()}
    cam, pic = {}, {}
    pic["width"] = p["image_width"]
    pic["height"] = p["image_height"]
    cam["chipsize"] = p["sensor_width"]
    cam["E"] = p["easting"]
    cam["N"] = p["northing"]
    cam["H"] = p["elevation"] - p["antenna_height"]
    cam["theta"] = np.radians(p["theta"])
    cam["phi"] = np.radians(p["phi"])
    cam["psi"] = np.radians(p["psi"])
    cam["sigma"] = (pic["width"] / cam["chipsize"]) * p["sigma"]
    pic["cropleft"] = p["crop_left"]
    pic["cropright"] = p["crop_right"]
    pic["croptop"] = p["crop_top"]
    pic["cropbottom"] = p["crop_bottom"]
    cam["H"] = cam["H"] - float(tide_of(stamp))
    return cam, pic


def synth_tracks(rng, n, nv, w, h):
    """(n, nv, 2) float32 tracks covering every branch of the filter: steady drift, stand-still (zero vectors),
    sharp turns, speed jumps, implausibly fast ones, exact repeats."""
    t = np.zeros((n, nv, 2), np.float64)
    t[:, 0, 0] = rng.uniform(20, w - 20, n)
    t[:, 0, 1] = rng.uniform(250, h - 20, n)      # below the horizon of the synthetic camera
    kind = rng.integers(0, 8, n)
    for i in range(n):
        step = rng.normal(0, 1.0, 2) * rng.choice([0.05, 0.4, 1.5])
        for k in range(1, nv):
            d = step + rng.normal(0, 0.05, 2)
            if kind[i] == 0:
                d = np.zeros(2)                                   # stand-still: speed 0, ratios x/0
            elif kind[i] == 1 and k == nv - 1:
                d = -step * rng.uniform(0.5, 2.0)                 # reversal: direction change
            elif kind[i] == 2 and k == nv - 1:
                d = step * rng.uniform(2.0, 6.0)                  # speed jump
            elif kind[i] == 3:
                d = step * 40.0                                   # far too fast
            elif kind[i] == 4 and k >= 2:
                d = step.copy()                                   # exactly repeated vector in pixels
            t[i, k] = t[i, k - 1] + d
    return t.astype(np.float32)


SCENARIOS = [
    # name, track_len T, interval s, files (stamps), filter (s2_cam_to_utm.py:84-88 defaults unless noted)
    dict(name="s0", T=2, dt=60, n=300, start="20190724-105500", step_s=120, files=6,
         filt=dict(max_speed=1.7, min_speed=0.0, max_speedfactor=2.5, max_angle=60, speed_threshold=0.1)),
    dict(name="s1", T=4, dt=30, n=200, start="20190725-235000", step_s=150, files=5,
         filt=dict(max_speed=1.2, min_speed=0.02, max_speedfactor=2.0, max_angle=45, speed_threshold=0.05)),
    dict(name="s2", T=9, dt=20, n=150, start="20190726-065700", step_s=180, files=3,
         filt=dict(max_speed=2.5, min_speed=0.01, max_speedfactor=3.0, max_angle=75, speed_threshold=0.08)),
]


def main():
    sys.path.insert(0, REF)
    sys.modules["shapefile"] = types.ModuleType("shapefile")   # placeholder, see the docstring
    import s2_cam_to_utm as s2
    ct = s2.ct
    real_camera = ct.Camera

    def fixed_camera(camname, date, paramfile_path, mask=0, tide_corr=0, tide_file="", datetime=""):
        cam = real_camera.__new__(real_camera)
        cam.date = dt.datetime.strptime(date, "%Y%m%d").date()
        cam.cam, cam.pic = camera_dicts(datetime)
        return cam

    ct.Camera = fixed_camera
    rng = np.random.default_rng(20240607)
    out = {}
    w = CALIB["image_width"] - CALIB["crop_left"] - CALIB["crop_right"]
    h = CALIB["image_height"] - CALIB["crop_top"] - CALIB["crop_bottom"]

    # (1) the projection alone, on scalar calls exactly as s2:249-251 makes them
    cam = fixed_camera("camX", "20190724", "", datetime="20190724-101500")
    pts = np.stack([rng.uniform(0, w, 500), rng.uniform(200, h, 500)], 1).astype(np.float32)
    res = []
    for x, y in pts.tolist():
        px, py = cam.photocords_cropped_to_uncropped(x, y)
        res.append(cam.photo_to_utm(px, py))
    out["calib_keys"] = np.array(sorted(CALIB))
    out["calib_values"] = np.array([float(CALIB[k]) for k in sorted(CALIB)], np.float64)
    out["p2u_stamp"] = np.array("20190724-101500")
    out["p2u_tide"] = np.array(tide_of("20190724-101500"))
    out["p2u_xy"] = pts
    out["p2u_utm"] = np.array(res, np.float64)

    # (2) whole folders through cam_to_utm
    for sc in SCENARIOS:
        with tempfile.TemporaryDirectory() as tmp:
            day = sc["start"].split("-")[0]
            src = os.path.join(tmp, day)
            dst = os.path.join(tmp, "utm")
            os.makedirs(src)
            os.makedirs(dst)
            t0 = dt.datetime.strptime(sc["start"], "%Y%m%d-%H%M%S")
            names = []
            for k in range(sc["files"]):
                stamp = (t0 + dt.timedelta(seconds=k * sc["step_s"])).strftime("%Y%m%d-%H%M%S")
                # s1_lucaskanade_tracking.py:394 naming
                name = "{}_{}sec_at_{}sec_tracks.npz".format(stamp, sc["T"] * sc["dt"], sc["dt"])
                tr = synth_tracks(rng, sc["n"], sc["T"] + 1, w, h)
                np.savez(os.path.join(src, name), tracks=tr, trackquality=np.zeros((sc["n"], sc["T"]), np.float32))
                out["%s_in_%02d_name" % (sc["name"], k)] = np.array(name)
                out["%s_in_%02d_tracks" % (sc["name"], k)] = tr
                out["%s_in_%02d_tide" % (sc["name"], k)] = np.array(tide_of(stamp))
                names.append(name)
            f = sc["filt"]
            args = (src, dst, "camX", f["max_speed"], f["min_speed"], f["max_speedfactor"], f["max_angle"],
                    f["speed_threshold"], "unused.xlsx", "unused.pickle")
            s2.cam_to_utm(args)
            outs = sorted(glob.glob(os.path.join(dst, "*.npz")))
            out["%s_n_in" % sc["name"]] = np.array(len(names))
            out["%s_n_out" % sc["name"]] = np.array(len(outs))
            out["%s_filter" % sc["name"]] = np.array([f["max_speed"], f["min_speed"], f["max_speedfactor"],
                                                      f["max_angle"], f["speed_threshold"]], np.float64)
            for k, path in enumerate(outs):
                z = np.load(path)
                out["%s_out_%02d_name" % (sc["name"], k)] = np.array(os.path.basename(path))
                for key in ("x", "y", "u", "v", "speed", "time"):
                    out["%s_out_%02d_%s" % (sc["name"], k, key)] = z[key]
                print(sc["name"], os.path.basename(path), {key: (z[key].shape, z[key].dtype) for key in z.files})

    # (3) the reference's failure mode: a single-vector track faster than speed_threshold -> max() of an empty list
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "20190724")
        os.makedirs(src)
        tr = np.array([[[100.0, 900.0], [100.1, 900.1]]], np.float32)   # 0.4 m/s: passes criterion 1
        np.savez(os.path.join(src, "20190724-120000_60sec_at_60sec_tracks.npz"), tracks=tr)
        try:
            s2.cam_to_utm((src, tmp, "camX", 1.7, 0.0, 2.5, 60, 0.0001, "u", "u"))
            out["t1_raises"] = np.array("")
        except ValueError as e:
            out["t1_raises"] = np.array(type(e).__name__)
        out["t1_tracks"] = tr
        out["t1_tide"] = np.array(tide_of("20190724-120000"))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(out), "arrays; numpy", np.__version__)


if __name__ == "__main__":
    main()
