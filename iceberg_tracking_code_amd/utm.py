"""Projection of finished tracks to map coordinates -- the consumer right after the tracking loop.

Host-side mirror of the reference's `s2_cam_to_utm.cam_to_utm` (s2_cam_to_utm.py:163-363) and of the part of
`imports.camtools.Camera` it uses (camtools.py:127-179 calibration arithmetic, 286-332 photo_to_utm, 414-421 crop
offsets).  The per-track work -- projection of every vertex, velocities, the three plausibility criteria -- runs in
the HIP kernel `k_project_tracks` (csrc/k_utm.hip) through `icelk_project_tracks` / `icelk_seg_project`; what stays
here is parameter preparation (fifteen trigonometric values per camera) and the hour bookkeeping of the output
files.  There is no CPU fallback.

Out of scope: reading the calibration workbook and the tide series -- `CameraModel` takes the numbers.
"""
import ctypes as C
import datetime as dt
import os

import numpy as np

from . import _lib
from .context import Context

# s2_cam_to_utm.py:84-88
REF_UTM_FILTER = dict(max_speed=1.7, min_speed=0.0, max_speedfactor=2.5, max_angle=60, speed_threshold=0.1)


class CameraStruct(C.Structure):
    """icelk_camera_t (include/icelk.h)."""
    _fields_ = [("X", C.c_double * 3), ("U", C.c_double * 3), ("V", C.c_double * 3), ("sigma", C.c_double),
                ("H", C.c_double), ("E", C.c_double), ("N", C.c_double), ("half_w", C.c_double),
                ("half_h", C.c_double), ("crop_left", C.c_double), ("crop_top", C.c_double)]


class FilterStruct(C.Structure):
    """icelk_utm_filter_t (include/icelk.h)."""
    _fields_ = [(k, C.c_double) for k in ("interval_s", "max_speed", "min_speed", "max_speedfactor", "max_angle",
                                          "speed_threshold")]


class CameraModel:
    """The `cam` / `pic` dictionaries of the reference's Camera (camtools.py:124-150), from explicit numbers.

    Arguments carry the names of the calibration workbook's columns (camtools.py:130-150).  `tide_elevation`, when
    given, lowers the camera height as camtools.py:179 does.  Values are kept as numpy scalars so that every later
    expression rounds as in the reference.
    """

    def __init__(self, image_width, image_height, sensor_width, easting, northing, elevation, antenna_height, theta,
                 phi, psi, sigma, crop_left=0, crop_right=0, crop_top=0, crop_bottom=0, tide_elevation=None):
        f = np.float64
        self.pic = dict(width=np.int64(image_width), height=np.int64(image_height), cropleft=np.int64(crop_left),
                        cropright=np.int64(crop_right), croptop=np.int64(crop_top), cropbottom=np.int64(crop_bottom))
        self.cam = dict(chipsize=f(sensor_width), E=f(easting), N=f(northing), H=f(elevation) - f(antenna_height),
                        theta=np.radians(f(theta)), phi=np.radians(f(phi)), psi=np.radians(f(psi)))
        self.cam["sigma"] = (self.pic["width"] / self.cam["chipsize"]) * f(sigma)
        if tide_elevation is not None:
            self.cam["H"] = self.cam["H"] - float(tide_elevation)

    def direction_vectors(self):
        """X (optical axis), U, V (image axes) in map coordinates -- Krimmel & Rasmussen eq. 7, with the products
        taken in the order of camtools.py:300-316."""
        th, ph, ps = self.cam["theta"], self.cam["phi"], self.cam["psi"]
        sth, cth, sph, cph, sps, cps = np.sin(th), np.cos(th), np.sin(ph), np.cos(ph), np.sin(ps), np.cos(ps)
        X = np.array([cth * cph, sth * cph, sph])
        U = np.array([sth * cps - cth * sph * sps, -cth * cps - sth * sph * sps, cph * sps])
        V = np.array([-sth * sps - cth * sph * cps, cth * sps - sth * sph * cps, cph * cps])
        return X, U, V

    def as_dict(self):
        """Fields of icelk_camera_t."""
        X, U, V = self.direction_vectors()
        return dict(X=X, U=U, V=V, sigma=self.cam["sigma"], H=self.cam["H"], E=self.cam["E"], N=self.cam["N"],
                    half_w=self.pic["width"] / 2.0, half_h=self.pic["height"] / 2.0,
                    crop_left=self.pic["cropleft"], crop_top=self.pic["croptop"])

    def struct(self):
        d = self.as_dict()
        s = CameraStruct()
        for k in ("X", "U", "V"):
            setattr(s, k, (C.c_double * 3)(*[float(v) for v in d[k]]))
        for k in ("sigma", "H", "E", "N", "half_w", "half_h", "crop_left", "crop_top"):
            setattr(s, k, float(d[k]))
        return s


def filter_struct(interval_s, max_speed, min_speed, max_speedfactor, max_angle, speed_threshold):
    return FilterStruct(float(interval_s), float(max_speed), float(min_speed), float(max_speedfactor),
                        float(max_angle), float(speed_threshold))


def _f64(a):
    return a.ctypes.data_as(_lib.f64p)


def _raise_like_reference(keep):
    if np.any(keep == 2):
        # s2_cam_to_utm.py:340 / :345 on a track with a single vector: max() of an empty list
        raise ValueError("max() arg is an empty sequence")


def project_tracks(ctx, tracks, camera, interval_s, max_speed=1.7, min_speed=0.0, max_speedfactor=2.5, max_angle=60,
                   speed_threshold=0.1):
    """`tracks` (n, n_vertices, 2) float32 -> dict(x, y, u, v, speed: (n, n_vertices-1) float64, keep: (n,) bool).

    x, y: map position of each vector's first vertex; u, v in m/s.  Raises ValueError where the reference does."""
    t = np.ascontiguousarray(tracks, dtype=np.float32)
    if t.ndim != 3 or t.shape[2] != 2:
        raise ValueError("tracks must be (n, n_vertices, 2)")
    n, nv = t.shape[0], t.shape[1]
    m = max(nv - 1, 0)
    out = {k: np.zeros((n, m), np.float64) for k in ("x", "y", "u", "v", "speed")}
    keep = np.zeros(n, np.uint8)
    cam, filt = camera.struct(), filter_struct(interval_s, max_speed, min_speed, max_speedfactor, max_angle,
                                               speed_threshold)
    ctx._ck(ctx._lib.icelk_project_tracks(ctx._h, t.ctypes.data_as(_lib.f32p), n, nv, C.byref(cam), C.byref(filt),
                                          _f64(out["x"]), _f64(out["y"]), _f64(out["u"]), _f64(out["v"]),
                                          _f64(out["speed"]), keep.ctypes.data_as(_lib.u8p)))
    _raise_like_reference(keep)
    out["keep"] = keep.astype(bool)
    return out


def project_segment(ctx, camera, interval_s, max_speed=1.7, min_speed=0.0, max_speedfactor=2.5, max_angle=60,
                    speed_threshold=0.1):
    """The same for the segment currently held on the device (SegmentTracker): only results cross PCIe."""
    n, _ = ctx.seg_live()
    nvec = C.c_int(0)
    nn = C.c_int(0)
    cam, filt = camera.struct(), filter_struct(interval_s, max_speed, min_speed, max_speedfactor, max_angle,
                                               speed_threshold)
    m = 16
    out = {k: np.zeros((n, m), np.float64) for k in ("x", "y", "u", "v", "speed")}
    keep = np.zeros(max(n, 1), np.uint8)
    ctx._ck(ctx._lib.icelk_seg_project(ctx._h, C.byref(cam), C.byref(filt), n, m, _f64(out["x"]), _f64(out["y"]),
                                       _f64(out["u"]), _f64(out["v"]), _f64(out["speed"]),
                                       keep.ctypes.data_as(_lib.u8p), C.byref(nn), C.byref(nvec)))
    keep = keep[:nn.value]
    _raise_like_reference(keep)
    out = {k: np.ascontiguousarray(v[:nn.value, :nvec.value]) for k, v in out.items()}
    out["keep"] = keep.astype(bool)
    return out


def _epoch(t):
    return int((t - dt.datetime(1970, 1, 1)).total_seconds())   # tracking_misc.py:237-239


def utm_name(label_time, tracking_interval):
    """Hourly output file name (s2_cam_to_utm.py:222-224, 359-361)."""
    return "{}_{}00_{}s_utm.npz".format(label_time.strftime("%Y%m%d"), label_time.strftime("%H"), tracking_interval)


def cam_to_utm(npz_paths, target_workspace, camera_for, max_speed=1.7, min_speed=0.0, max_speedfactor=2.5,
               max_angle=60, speed_threshold=0.1, ctx=None, save=True):
    """One day's folder of track files -> hourly velocity files, as s2_cam_to_utm.cam_to_utm (s2:163-363).

    `npz_paths`: the folder's `*_tracks.npz` (any order; sorted here as s2:177 does).  `camera_for(stamp)` returns
    the CameraModel for a file's '%Y%m%d-%H%M%S' stamp (the reference builds one Camera per file with the tide of
    that minute, s2:236-237).  Returns [(file name, dict(x, y, u, v, speed, time))] in writing order; with `save`
    the files are written with np.savez exactly as the reference does."""
    paths = sorted(npz_paths)
    if not paths:
        return []
    own = ctx is None
    if own:
        ctx = Context(64, 64, n_slots=1, max_pts=1 << 18)
    try:
        interval = int(os.path.basename(paths[0]).split("_")[-2].split("sec")[0])   # s2:182
        keys = ("x", "y", "u", "v", "speed", "time")
        cur = {k: [] for k in keys}
        nxt = {k: [] for k in keys}
        written = []

        def flush(label):
            name = utm_name(label, interval)
            arrays = {k: (np.concatenate(cur[k]) if cur[k] else np.array([])) for k in keys}
            if save:
                np.savez(os.path.join(target_workspace, name), **arrays)
            written.append((name, arrays))

        next_hour = None
        t_file = None
        for c, path in enumerate(paths):
            stamp = os.path.basename(path).split("_")[0]
            t_file = dt.datetime.strptime(stamp, "%Y%m%d-%H%M%S")
            current_hour = t_file.hour
            if c == 0:
                next_hour = (t_file + dt.timedelta(hours=1)).hour
            if current_hour == next_hour:                             # s2:214-241: a new hour starts
                flush(t_file - dt.timedelta(hours=1))
                cur, nxt = nxt, {k: [] for k in keys}
                next_hour = (t_file + dt.timedelta(hours=1)).hour
            tracks = np.load(path)["tracks"]
            if tracks.size == 0:
                continue
            r = project_tracks(ctx, tracks, camera_for(stamp), interval, max_speed, min_speed, max_speedfactor,
                               max_angle, speed_threshold)
            m = tracks.shape[1] - 1
            times = [t_file + dt.timedelta(seconds=(i - 1) * interval) for i in range(1, m + 1)]     # s2:285
            in_cur = np.array([t.hour == current_hour for t in times], bool)                         # s2:294
            epochs = np.array([_epoch(t) for t in times], np.int64)
            kept = r["keep"]
            nk = int(kept.sum())
            for sel, dst in ((in_cur, cur), (~in_cur, nxt)):
                if nk == 0 or not sel.any():
                    continue
                for k in ("x", "y", "u", "v", "speed"):
                    dst[k].append(r[k][kept][:, sel].ravel())          # track-major, as the per-track extend
                dst["time"].append(np.tile(epochs[sel], nk))
        flush(t_file)                                                 # s2:358-361
        return written
    finally:
        if own:
            ctx.close()
