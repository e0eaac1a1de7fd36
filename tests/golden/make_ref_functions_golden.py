#!/usr/bin/env python3
"""Generates tests/golden/ref_functions_golden.npz by RUNNING TWO MORE FUNCTIONS OF THE REFERENCE (development container only):

(a) `imports.camtools.Camera.__init__` (camtools.py:111-179) itself -- round 1 filled the `cam` / `pic` dictionaries of a
    `Camera.__new__` instance by hand -- with tide correction, for three time stamps;
(b) `s3_utm_to_gridded_utm.utm_to_gridded_utm` (s3:222-446, plot_switch 0) itself -- round 1 restated its loop body
    s3:391-421 inside the generator -- on one synthetic day of two cameras, one of them with a clock drift, 30-minute windows.

What these functions READ is supplied, not emulated: both call `pandas.read_excel` on the calibration / clock-drift
workbooks (openpyxl is not in the image; the reference ships no workbook either), so `pandas.read_excel` is pointed at
the in-memory tables below for the duration of the run -- tables are inputs, no arithmetic of the reference is replaced;
the tide series `Camera.__init__` unpickles is a DataFrame this script pickles itself; `shapefile` (pyshp, imported at
camtools.py:16 and unused here) is an empty placeholder module as in make_utm_golden.py.
For (b) the generator also records, per output file, the velocities the function had selected for that window -- by
calling the reference's own `trm.correct_time_drift` / `trm.return_velocities_by_time` with the arguments the function
forms (s3:305-322) -- because that concatenated array is what `gridding.bin_velocities` takes (the day / camera /
window bookkeeping itself is out of scope).
Committed: this script and the data; no reference source.
"""
import datetime as dt
import glob
import os
import sys
import tempfile
import types

import numpy as np
import pandas as pd

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_functions_golden.npz")

CALIB = dict(camera="cam1", start_day=20190701, end_day=20190831, start_time="10:00", tracking_duration=2.0,
             image_width=3456, image_height=2304, sensor_width=22.3, easting=497812.37, northing=6521034.81,
             elevation=431.62, antenna_height=1.35, theta=201.4, phi=-11.85, psi=1.27, sigma=24.6,
             crop_left=12, crop_right=20, crop_top=1000, crop_bottom=4, mask="none.shp")
CALIB2 = dict(CALIB, camera="cam2", start_time="10:30", tracking_duration=1.5, easting=498950.1, theta=158.2)
DRIFT = [dict(cam="cam2", start_date=20190720, end_date=20190731, drift_start_sec=37.5, drift_pday_sec=1.2)]


def main():
    sys.path.insert(0, REF)
    sys.modules["shapefile"] = types.ModuleType("shapefile")
    import imports.camtools as ct
    import imports.tracking_misc as trm
    import s3_utm_to_gridded_utm as s3
    tables = {}
    real_read_excel = pd.read_excel
    pd.read_excel = lambda path, *a, **k: tables[os.path.basename(str(path))].copy()
    out = {}
    rng = np.random.default_rng(31)
    try:
        with tempfile.TemporaryDirectory() as tmp:
            # ---- (a) Camera.__init__ with tide correction --------------------------------------------------------
            tables["parameter_file.xlsx"] = pd.DataFrame([CALIB, CALIB2])
            os.makedirs(os.path.join(tmp, "data"))
            minutes = pd.date_range("2019-07-24 00:00", "2019-07-25 23:59", freq="min")
            tide = (1.8 * np.sin(2 * np.pi * np.arange(len(minutes)) / 745.0) + 0.3).astype(np.float32)
            pd.DataFrame(dict(date=minutes, depth_tide_ellipsoid=tide)).to_pickle(os.path.join(tmp, "data", "tides.pickle"))
            stamps = ["20190724-101500", "20190724-235959", "20190725-000000"]
            keys_cam = ("chipsize", "E", "N", "H", "theta", "phi", "psi", "sigma")
            keys_pic = ("width", "height", "cropleft", "cropright", "croptop", "cropbottom")
            rows = []
            for camname, stamp in (("cam1", stamps[0]), ("cam2", stamps[1]), ("cam1", stamps[2])):
                c = ct.Camera(camname, stamp.split("-")[0], os.path.join(tmp, "parameter_file.xlsx"), tide_corr=1,
                              tide_file="tides.pickle", datetime=stamp)
                t = dt.datetime.strptime(stamp, "%Y%m%d-%H%M%S").replace(second=0)
                rows.append([float(c.cam[k]) for k in keys_cam] + [float(c.pic[k]) for k in keys_pic] +
                            [float(tide[list(minutes).index(pd.Timestamp(t))])])
                # one projection through the instance the constructor made
                px, py = c.photocords_cropped_to_uncropped(1234.5, 321.25)
                rows[-1] += list(c.photo_to_utm(px, py))
            out["cam_names"] = np.array(["cam1", "cam2", "cam1"])
            out["cam_stamps"] = np.array(stamps)
            out["cam_fields"] = np.array(list(keys_cam) + list(keys_pic) + ["tide", "utm_x", "utm_y"])
            out["cam_values"] = np.array(rows, np.float64)
            calib_cols = [k for k in CALIB if k not in ("camera", "start_time", "mask")]
            out["calib_cols"] = np.array(calib_cols)
            out["calib_rows"] = np.array([[float(r[k]) for k in calib_cols] for r in (CALIB, CALIB2)], np.float64)

            # ---- (b) utm_to_gridded_utm -------------------------------------------------------------------------
            tables["camera_time_drifts.xlsx"] = pd.DataFrame(DRIFT)
            ang = np.sort(rng.uniform(0, 2 * np.pi, 36))
            rad = rng.uniform(900, 2000, 36)
            fjord = dict(x=497000.0 + np.round(1.5 * rad * np.cos(ang), 1), y=6521000.0 + np.round(rad * np.sin(ang), 1))
            np.savez(os.path.join(tmp, "fjord_outline.npz"), **fjord)
            day = dt.datetime(2019, 7, 24)
            spacing, thr = 250, 8
            epoch = lambda t: trm.datetime_to_epoch(t)   # noqa: E731
            head = os.path.join(tmp, "out")
            for camname, hours, n in (("cam1", (10, 11), 6000), ("cam2", (10, 11, 12), 4000)):
                ws = os.path.join(head, camname, "utm")
                os.makedirs(ws)
                for hr in hours:
                    t0 = day + dt.timedelta(hours=hr)
                    tt = np.sort(rng.uniform(epoch(t0), epoch(t0 + dt.timedelta(hours=1)), n))
                    x = rng.uniform(min(fjord["x"]) - 200, max(fjord["x"]) + 200, n)
                    y = rng.uniform(min(fjord["y"]) - 200, max(fjord["y"]) + 200, n)
                    k = n // 10
                    x[:k] = min(fjord["x"]) + spacing * rng.integers(0, 20, k)            # exactly on cell edges
                    y[k:2 * k] = max(fjord["y"]) - spacing * rng.integers(0, 15, k)
                    u = rng.normal(0.1, 0.3, n) * 10.0 ** rng.integers(-3, 2, n)
                    v = rng.normal(-0.05, 0.2, n) * 10.0 ** rng.integers(-3, 2, n)
                    np.savez(os.path.join(ws, t0.strftime("%Y%m%d_%H00") + "-%02d00_60sec_utm.npz" % (hr + 1)),
                             x=x, y=y, u=u, v=v, speed=np.hypot(u, v), time=tt)
            target = os.path.join(tmp, "gridded")
            os.makedirs(target)
            args = (["cam1", "cam2"], head, "utm", target, tmp, os.path.join(tmp, "parameter_file.xlsx"),
                    os.path.join(tmp, "camera_time_drifts.xlsx"), os.path.join(tmp, "fjord_outline.npz"), pd.Timestamp(day),
                    30 / 60.0, spacing, 0.5, thr, 0)
            s3.utm_to_gridded_utm(args)
            files = sorted(glob.glob(os.path.join(target, "*.npz")))
            out["grid_fjord_x"], out["grid_fjord_y"] = fjord["x"], fjord["y"]
            out["grid_spacing"], out["grid_threshold"] = np.array(spacing), np.array(thr)
            out["grid_n_files"] = np.array(len(files))
            drift_file = tables["camera_time_drifts.xlsx"]
            for fi, path in enumerate(files):
                z = np.load(path, allow_pickle=True)
                name = os.path.basename(path)
                out["grid_%02d_name" % fi] = np.array(name)
                for key in ("grid_id", "i", "j", "x", "y", "u", "v", "speed", "count", "rows", "cols", "topleft"):
                    out["grid_%02d_%s" % (fi, key)] = np.asarray(z[key])
                # the velocities the function had selected for this window (s3:299-353), by the reference's own helpers
                start = dt.datetime.strptime(name[:13], "%Y%m%d_%H%M")
                end = start + dt.timedelta(minutes=int(name.split("_")[2].split("min")[0]))
                sel = []
                for camname in ("cam1", "cam2"):
                    try:
                        corr = trm.correct_time_drift(camname, "20190724", drift_file)
                    except Exception:
                        corr = 0
                    ws = os.path.join(head, camname, "utm")
                    r = trm.return_velocities_by_time(ws, start - dt.timedelta(seconds=corr), end - dt.timedelta(seconds=corr))
                    if len(r[2]) > 0:
                        sel.append(r)
                for k, key in enumerate(("px", "py", "pu", "pv")):
                    out["grid_%02d_%s" % (fi, key)] = np.concatenate([s[k] for s in sel])
                print(name, "cells measured", len(z["grid_id"]), "velocities", len(out["grid_%02d_px" % fi]))
    finally:
        pd.read_excel = real_read_excel
    np.savez_compressed(OUT, **out)
    import matplotlib
    print("wrote", OUT, os.path.getsize(OUT), "bytes; numpy", np.__version__, "pandas", pd.__version__, "matplotlib", matplotlib.__version__)


if __name__ == "__main__":
    main()
