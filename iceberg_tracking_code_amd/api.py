"""cv2-shaped entry points: the three functions the reference's frame loop calls.

    cvtColor(frame, COLOR_BGR2GRAY)                       s1_lucaskanade_tracking.py:283,311
    calcOpticalFlowPyrLK(img0, img1, p0, None, **lk)      s1_lucaskanade_tracking.py:323,326
    goodFeaturesToTrack(gray, mask=mask, **feature)       s1_lucaskanade_tracking.py:437

Same names, argument meaning, return shapes and None/empty behaviour as cv2, so the reference loop runs
with `import iceberg_tracking_code_amd as cv2`.  Every call goes to the HIP library
through a process-wide default Context; each call uploads its images, so this surface is PCIe-bound --
the device-resident loop is tracker.SegmentTracker.
"""
import numpy as np

from .context import (Context, DEFAULT_CRITERIA, GRAY_CV3, GRAY_CV4, OPTFLOW_LK_GET_MIN_EIGENVALS,  # noqa: F401
                      OPTFLOW_USE_INITIAL_FLOW, TERM_CRITERIA_COUNT, TERM_CRITERIA_EPS, TERM_CRITERIA_MAX_ITER)

COLOR_BGR2GRAY = 6
COLOR_RGB2GRAY = 7

_default = {"ctx": None, "gray_variant": GRAY_CV4, "device": 0, "variants": {}}


def set_gray_variant(variant):
    """4 = OpenCV 4.x 15-bit coefficients (default: the reference's environment.yml:254 pins opencv 4.9.0),
    3 = OpenCV 3.x 14-bit (the "3.1.0" its README.md:8 mentions)."""
    if variant not in (GRAY_CV3, GRAY_CV4):
        raise ValueError("variant must be 3 or 4")
    _default["gray_variant"] = variant


def set_variant(name, value):
    """A named build-dependent variant of OpenCV's arithmetic for the cv2-shaped calls of this module: "lk_sums" 0 | 1 | 2
    (exact / OpenCV 3.x SSE2 / 4.x CV_SIMD128 summation order of the LK sums), "sobel_fma" 0..3, "eig_fma" 0 | 1
    (icelk_set_variant; DESIGN.md section 2 has how far apart they are).  0 = default."""
    if name not in ("lk_sums", "sobel_fma", "eig_fma"):
        raise ValueError("unknown variant %r" % (name,))
    _default["variants"][name] = int(value)
    if _default["ctx"] is not None:
        _default["ctx"].set_variant(name, value)


def set_device(device):
    if _default["ctx"] is not None and _default["device"] != device:
        _default["ctx"].close()
        _default["ctx"] = None
    _default["device"] = int(device)


def default_context(w, h, n_pts=0):
    """Process-wide context, re-created when a larger frame or point count shows up."""
    ctx = _default["ctx"]
    need_pts = max(int(n_pts), 1 << 18)
    if ctx is None or w > ctx.max_w or h > ctx.max_h or need_pts > ctx.max_pts:
        mw = max(w, ctx.max_w if ctx else 0)
        mh = max(h, ctx.max_h if ctx else 0)
        mp = max(need_pts, ctx.max_pts if ctx else 0)
        if ctx is not None:
            ctx.close()
        ctx = Context(mw, mh, n_slots=2, max_pts=mp, device=_default["device"])
        for name, value in _default["variants"].items():
            ctx.set_variant(name, value)
        _default["ctx"] = ctx
    return ctx


def release():
    if _default["ctx"] is not None:
        _default["ctx"].close()
        _default["ctx"] = None


def cvtColor(src, code):
    a = np.asarray(src)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("cvtColor: only 8-bit 3-channel input is supported on this path")
    if code == COLOR_RGB2GRAY:
        a = a[:, :, ::-1]
    elif code != COLOR_BGR2GRAY:
        raise ValueError("cvtColor: only COLOR_BGR2GRAY / COLOR_RGB2GRAY are part of this path")
    h, w = a.shape[:2]
    ctx = default_context(w, h)
    ctx.upload_bgr(0, a, _default["gray_variant"])
    return ctx.download_level(0, 0)


def _check_gray(img, name):
    a = np.asarray(img)
    if a.dtype != np.uint8 or a.ndim != 2:
        raise ValueError("%s must be an 8-bit single-channel image" % name)
    return a


def calcOpticalFlowPyrLK(prevImg, nextImg, prevPts, nextPts=None, status=None, err=None, winSize=(21, 21), maxLevel=3,
                         criteria=DEFAULT_CRITERIA, flags=0, minEigThreshold=1e-4):
    """-> (nextPts (N,1,2) float32, status (N,1) uint8, err (N,1) float32), like cv2."""
    i0, i1 = _check_gray(prevImg, "prevImg"), _check_gray(nextImg, "nextImg")
    if i0.shape != i1.shape:
        raise ValueError("prevImg and nextImg differ in size")
    p0 = np.asarray(prevPts, dtype=np.float32)
    if p0.size % 2:
        raise ValueError("prevPts must hold (x, y) pairs")
    n = p0.size // 2
    if n == 0:
        return (np.zeros((0, 1, 2), np.float32), np.zeros((0, 1), np.uint8), np.zeros((0, 1), np.float32))
    h, w = i0.shape
    ctx = default_context(w, h, n)
    ctx.upload_gray(0, i0)
    ctx.upload_gray(1, i1)
    return ctx.pyrlk(0, 1, p0, nextPts, winSize, maxLevel, criteria, flags, minEigThreshold)


def goodFeaturesToTrack(image, maxCorners, qualityLevel, minDistance, corners=None, mask=None, blockSize=3,
                        useHarrisDetector=False, k=0.04):
    """-> (M,1,2) float32, or None when nothing passes (the reference guards this at s1:445)."""
    if useHarrisDetector:
        raise ValueError("useHarrisDetector=True is not part of this path (the reference uses Shi-Tomasi)")
    img = _check_gray(image, "image")
    if not (qualityLevel > 0) or minDistance < 0 or maxCorners < 0:
        raise ValueError("goodFeaturesToTrack: qualityLevel > 0, minDistance >= 0, maxCorners >= 0 required")
    h, w = img.shape
    ctx = default_context(w, h)
    ctx.upload_gray(0, img)
    if mask is not None:
        m = _check_gray(mask, "mask")
        if m.shape != img.shape:
            raise ValueError("mask size differs from the image")
        ctx.set_mask(m)
    return ctx.good_features(0, maxCorners, qualityLevel, minDistance, use_mask=mask is not None, blockSize=blockSize)
