"""ctypes front end of oracle/libicelk_oracle.so (built from icelk_oracle.c).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Argument shapes follow the cv2 calls the
reference makes (s1_lucaskanade_tracking.py:311,323,326,437).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libicelk_oracle.so")

CRIT_COUNT = 1
CRIT_EPS = 2
FLAG_INITIAL_FLOW = 4
FLAG_MIN_EIGENVALS = 8

_lib = None
_u8p = C.POINTER(C.c_uint8)
_f32p = C.POINTER(C.c_float)
_i16p = C.POINTER(C.c_int16)
_i32p = C.POINTER(C.c_int)
_f64p = C.POINTER(C.c_double)


class UtmCamera(C.Structure):
    """orc_camera_t (utm_oracle.c): direction cosines as camtools.py:300-316 forms them, plus the scalars."""
    _fields_ = [("X", C.c_double * 3), ("U", C.c_double * 3), ("V", C.c_double * 3), ("sigma", C.c_double),
                ("H", C.c_double), ("E", C.c_double), ("N", C.c_double), ("half_w", C.c_double),
                ("half_h", C.c_double), ("crop_left", C.c_double), ("crop_top", C.c_double)]


class UtmFilter(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("interval_s", "max_speed", "min_speed", "max_speedfactor", "max_angle",
                                          "speed_threshold")]


def build(force=False):
    """Compile the C restatement with the committed Makefile (gcc only)."""
    srcs = [os.path.join(_HERE, f) for f in ("icelk_oracle.c", "utm_oracle.c", "mask_oracle.c", "grid_oracle.c")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_bgr2gray.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int, C.c_int]
        L.orc_pyrdown.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int]
        L.orc_pyramid_levels.argtypes = [C.c_int] * 5
        L.orc_build_pyramid.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _u8p, _i32p]
        L.orc_scharr.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _i16p, C.c_int]
        L.orc_pyrlk.argtypes = [_u8p, C.c_int, _u8p, C.c_int, C.c_int, C.c_int, _f32p, _f32p, _u8p, _f32p,
                                C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                                C.c_double]
        L.orc_track_fb.argtypes = [_u8p, C.c_int, _u8p, C.c_int, C.c_int, C.c_int, _f32p, _f32p, _f32p, _u8p,
                                   _u8p, _f32p, _f32p, _f32p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_double, C.c_double, C.c_float]
        L.orc_min_eig_map.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p]
        L.orc_set_fb_distance.argtypes = [C.c_int]
        L.orc_set_variant.argtypes = [C.c_char_p, C.c_int]
        L.orc_get_variant.argtypes = [C.c_char_p]
        L.orc_fb_distance.argtypes = [C.c_float] * 4
        L.orc_fb_distance.restype = C.c_float
        L.orc_good_features.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int, C.c_int, C.c_double,
                                        C.c_double, C.c_int, _f32p, C.c_int, _i32p]
        L.orc_project_tracks.argtypes = [_f32p, C.c_int, C.c_int, C.POINTER(UtmCamera), C.POINTER(UtmFilter), _f64p,
                                         _f64p, _f64p, _f64p, _f64p, _u8p]
        L.orc_points_in_polygon.argtypes = [_f64p, C.c_int, _f64p, C.c_int, _u8p]
        L.orc_grid_bin.argtypes = [_f64p, _f64p, _f64p, _f64p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int,
                                   C.c_int, _u8p, _i32p, _f64p, _f64p, _f64p]
        L.orc_polygon_mask.argtypes = [_f64p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, _u8p, C.c_int]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def _gray(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim != 2:
        raise ValueError("expected HxW uint8 image")
    return a


def _chk(rc):
    if rc < 0:
        raise RuntimeError("oracle error %d" % rc)
    return rc


def lk_stats(reset=True):
    """(template patches built, LK iterations, calls) since the last reset."""
    out = (C.c_longlong * 3)()
    lib().orc_lk_stats(out, 1 if reset else 0)
    return tuple(int(v) for v in out)


def set_threads(n):
    return lib().orc_set_threads(int(n))


def set_fb_distance(form):
    """0 = np.hypot on float32 (s1:330, default), 1 = (dx**2 + dy**2)**0.5 in float32 (s0_1:99)."""
    lib().orc_set_fb_distance(int(form))


def fb_distance(p0, p0r):
    """dist of s1:329-330 for (n, 2) float32 point arrays, one scalar call per point (for the known-answer test)."""
    p0 = np.asarray(p0, np.float32).reshape(-1, 2)
    p0r = np.asarray(p0r, np.float32).reshape(-1, 2)
    f = lib().orc_fb_distance
    return np.array([f(a[0], a[1], b[0], b[1]) for a, b in zip(p0, p0r)], np.float32)


def bgr2gray(src, variant=3):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    h, w, c = src.shape
    assert c == 3
    dst = np.empty((h, w), np.uint8)
    _chk(lib().orc_bgr2gray(_p(src, _u8p), w, h, 3 * w, _p(dst, _u8p), w, variant))
    return dst


def pyrdown(img):
    img = _gray(img)
    h, w = img.shape
    dst = np.empty(((h + 1) // 2, (w + 1) // 2), np.uint8)
    _chk(lib().orc_pyrdown(_p(img, _u8p), w, h, w, _p(dst, _u8p), dst.shape[1]))
    return dst


def pyramid_levels(w, h, win, max_level):
    return lib().orc_pyramid_levels(w, h, win[0], win[1], max_level)


def build_pyramid(img, win=(21, 21), max_level=3):
    """Returns the list of level images that buildOpticalFlowPyramid would hold (unpadded)."""
    img = _gray(img)
    h, w = img.shape
    out = np.empty(2 * w * h, np.uint8)
    nlev = C.c_int(0)
    _chk(lib().orc_build_pyramid(_p(img, _u8p), w, h, w, win[0], win[1], max_level, _p(out, _u8p), C.byref(nlev)))
    levels, off = [], 0
    lw, lh = w, h
    for _ in range(nlev.value + 1):
        levels.append(out[off:off + lw * lh].reshape(lh, lw).copy())
        off += lw * lh
        lw, lh = (lw + 1) // 2, (lh + 1) // 2
    return levels


def scharr(img):
    img = _gray(img)
    h, w = img.shape
    dst = np.empty((h, w, 2), np.int16)
    _chk(lib().orc_scharr(_p(img, _u8p), w, h, w, _p(dst, _i16p), 2 * w))
    return dst


def _crit(criteria):
    t, cnt, eps = criteria
    return int(t), int(cnt), float(eps)


def pyrlk(prev, nxt, prev_pts, next_pts=None, winSize=(21, 21), maxLevel=3,
          criteria=(CRIT_COUNT | CRIT_EPS, 30, 0.01), flags=0, minEigThreshold=1e-4):
    """cv2.calcOpticalFlowPyrLK-shaped: returns (nextPts (N,1,2) f32, status (N,1) u8, err (N,1) f32)."""
    prev, nxt = _gray(prev), _gray(nxt)
    assert prev.shape == nxt.shape
    h, w = prev.shape
    p0 = np.ascontiguousarray(prev_pts, dtype=np.float32).reshape(-1, 2)
    n = p0.shape[0]
    if flags & FLAG_INITIAL_FLOW:
        p1 = np.ascontiguousarray(next_pts, dtype=np.float32).reshape(-1, 2).copy()
    else:
        p1 = np.zeros((n, 2), np.float32)
    st = np.zeros(n, np.uint8)
    er = np.zeros(n, np.float32)
    t, cnt, eps = _crit(criteria)
    _chk(lib().orc_pyrlk(_p(prev, _u8p), w, _p(nxt, _u8p), w, w, h, _p(p0, _f32p), _p(p1, _f32p), _p(st, _u8p),
                         _p(er, _f32p), n, winSize[0], winSize[1], maxLevel, t, cnt, eps, flags, minEigThreshold))
    return p1.reshape(-1, 1, 2), st.reshape(-1, 1), er.reshape(-1, 1)


def track_fb(img0, img1, p0, winSize=(21, 21), maxLevel=3, criteria=(CRIT_COUNT | CRIT_EPS, 30, 0.01),
             minEigThreshold=1e-4, fb_threshold=1.0):
    """The reference's forward + backward + distance test (s1:323-333) in one call."""
    img0, img1 = _gray(img0), _gray(img1)
    h, w = img0.shape
    p0 = np.ascontiguousarray(p0, dtype=np.float32).reshape(-1, 2)
    n = p0.shape[0]
    p1 = np.zeros((n, 2), np.float32)
    p0r = np.zeros((n, 2), np.float32)
    st_f = np.zeros(n, np.uint8)
    st_b = np.zeros(n, np.uint8)
    er_f = np.zeros(n, np.float32)
    er_b = np.zeros(n, np.float32)
    dist = np.zeros(n, np.float32)
    valid = np.zeros(n, np.uint8)
    t, cnt, eps = _crit(criteria)
    _chk(lib().orc_track_fb(_p(img0, _u8p), w, _p(img1, _u8p), w, w, h, _p(p0, _f32p), _p(p1, _f32p),
                            _p(p0r, _f32p), _p(st_f, _u8p), _p(st_b, _u8p), _p(er_f, _f32p), _p(er_b, _f32p),
                            _p(dist, _f32p), _p(valid, _u8p), n, winSize[0], winSize[1], maxLevel, t, cnt, eps,
                            minEigThreshold, fb_threshold))
    return dict(p1=p1, p0r=p0r, st_fwd=st_f, st_bwd=st_b, err_fwd=er_f, err_bwd=er_b, dist=dist, valid=valid)


VARIANTS = {"lk_sums": (0, 1, 2), "sobel_fma": (0, 1, 2, 3), "eig_fma": (0, 1)}


def set_variant(name, value):
    """Select a named variant of a build-dependent OpenCV semantic (icelk_oracle.c: orc_set_variant); 0 = the default,
    which is what the HIP kernels compute.  Process-wide: reset it when done (`variants()` does)."""
    if lib().orc_set_variant(name.encode(), int(value)) != 0:
        raise ValueError("unknown oracle variant %s=%r" % (name, value))


def get_variant(name):
    return lib().orc_get_variant(name.encode())


class variants:
    """with oracle.variants(lk_sums=1, eig_fma=1): ...   -- the switches go back to their defaults on exit."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        for k, v in self.kw.items():
            set_variant(k, v)
        return self

    def __exit__(self, *exc):
        for k in VARIANTS:
            set_variant(k, 0)


def min_eig_map(img, blockSize=3):
    img = _gray(img)
    h, w = img.shape
    eig = np.empty((h, w), np.float32)
    _chk(lib().orc_min_eig_map(_p(img, _u8p), w, h, w, blockSize, _p(eig, _f32p)))
    return eig


def good_features(img, maxCorners, qualityLevel, minDistance, mask=None, blockSize=3):
    """cv2.goodFeaturesToTrack-shaped: (M,1,2) float32, or None when nothing is found."""
    img = _gray(img)
    h, w = img.shape
    cap = w * h if maxCorners <= 0 else int(maxCorners)
    cap = min(cap, w * h)
    out = np.empty((max(cap, 1), 2), np.float32)
    n = C.c_int(0)
    if mask is not None:
        mask = _gray(mask)
        assert mask.shape == img.shape
        mp, ms = _p(mask, _u8p), w
    else:
        mp, ms = None, 0
    _chk(lib().orc_good_features(_p(img, _u8p), w, h, w, mp, ms, int(maxCorners), float(qualityLevel),
                                 float(minDistance), int(blockSize), _p(out, _f32p), cap, C.byref(n)))
    if n.value == 0:
        return None
    return out[:n.value].reshape(-1, 1, 2).copy()


def project_tracks(tracks, cam, filt):
    """utm_oracle.c: `tracks` (n, nv, 2) f32; `cam` / `filt`: dicts with the fields of UtmCamera / UtmFilter.
    Returns dict(x, y, u, v, speed: (n, nv-1) f64; keep: (n,) u8 -- 1 kept, 0 dropped, 2 reference raises)."""
    t = np.ascontiguousarray(tracks, dtype=np.float32)
    n, nv, two = t.shape
    assert two == 2
    c = UtmCamera()
    for k in ("X", "U", "V"):
        setattr(c, k, (C.c_double * 3)(*[float(v) for v in cam[k]]))
    for k in ("sigma", "H", "E", "N", "half_w", "half_h", "crop_left", "crop_top"):
        setattr(c, k, float(cam[k]))
    f = UtmFilter(**{k: float(filt[k]) for k, _ in UtmFilter._fields_})
    m = nv - 1
    out = {k: np.zeros((n, m), np.float64) for k in ("x", "y", "u", "v", "speed")}
    out["keep"] = np.zeros(n, np.uint8)
    _chk(lib().orc_project_tracks(_p(t, _f32p), n, nv, C.byref(c), C.byref(f), _p(out["x"], _f64p),
                                  _p(out["y"], _f64p), _p(out["u"], _f64p), _p(out["v"], _f64p),
                                  _p(out["speed"], _f64p), _p(out["keep"], _u8p)))
    return out


def polygon_mask(poly, crop_left, crop_top, w, h):
    """mask_oracle.c: polygon (n, 2) on the uncropped photo -> (h, w) u8 mask (255 inside) of the cropped frame."""
    p = np.ascontiguousarray(poly, dtype=np.float64).reshape(-1, 2)
    m = np.zeros((h, w), np.uint8)
    _chk(lib().orc_polygon_mask(_p(p, _f64p), len(p), float(crop_left), float(crop_top), w, h, _p(m, _u8p), w))
    return m


def points_in_polygon(poly, pts):
    """grid_oracle.c: matplotlib's Path(poly).contains_points(pts) (radius 0) -> bool array."""
    p = np.ascontiguousarray(poly, dtype=np.float64).reshape(-1, 2)
    q = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 2)
    out = np.zeros(len(q), np.uint8)
    _chk(lib().orc_points_in_polygon(_p(p, _f64p), len(p), _p(q, _f64p), len(q), _p(out, _u8p)))
    return out.astype(bool)


def grid_bin(x, y, u, v, left, top, spacing, cols, rows, cell_on):
    """grid_oracle.c: per cell (index i * rows + j) count, mean_u, mean_v, speed of the velocities inside it."""
    a = [np.ascontiguousarray(t, dtype=np.float64).ravel() for t in (x, y, u, v)]
    on = np.ascontiguousarray(cell_on, dtype=np.uint8).ravel()
    nc = cols * rows
    assert on.size == nc
    cnt = np.zeros(nc, np.int32)
    mu, mv, sp = (np.zeros(nc, np.float64) for _ in range(3))
    _chk(lib().orc_grid_bin(_p(a[0], _f64p), _p(a[1], _f64p), _p(a[2], _f64p), _p(a[3], _f64p), len(a[0]), float(left),
                            float(top), float(spacing), cols, rows, _p(on, _u8p), _p(cnt, _i32p), _p(mu, _f64p),
                            _p(mv, _f64p), _p(sp, _f64p)))
    return dict(count=cnt, mean_u=mu, mean_v=mv, speed=sp)
