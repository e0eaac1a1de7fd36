"""Context: one GPU handle (icelk_t) with numpy-facing methods.

This is the host side of the hot path: everything here is argument marshalling around the C ABI of
include/icelk.h.  One Context per process per GPU (see DESIGN.md "Multi-GPU").
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, f32p, u8p

TERM_CRITERIA_COUNT = 1
TERM_CRITERIA_MAX_ITER = 1
TERM_CRITERIA_EPS = 2
OPTFLOW_USE_INITIAL_FLOW = 4
OPTFLOW_LK_GET_MIN_EIGENVALS = 8
GRAY_CV3 = 3
GRAY_CV4 = 4
LK_GENERIC_KERNEL = 0x100
LK_MULTI_PER_WAVE = 0x200
FB_HYPOT = 0
FB_SQRT = 1

DEFAULT_CRITERIA = (TERM_CRITERIA_COUNT | TERM_CRITERIA_EPS, 30, 0.01)


def _u8(a):
    return a.ctypes.data_as(u8p)


def _f32(a):
    return a.ctypes.data_as(f32p)


def _gray2d(img, name="image"):
    a = np.asarray(img)
    if a.dtype != np.uint8 or a.ndim != 2:
        raise ValueError("%s must be a 2-D uint8 array (got %s %s)" % (name, a.dtype, a.shape))
    if a.strides[1] != 1 or a.strides[0] < a.shape[1]:
        a = np.ascontiguousarray(a)
    return a


def _criteria(criteria):
    t, cnt, eps = criteria
    return int(t), int(cnt), float(eps)


class Context:
    """Owns the device memory of `n_slots` resident frames (+ pyramids) and all point buffers."""

    def __init__(self, max_w, max_h, n_slots=3, max_pts=1 << 18, device=0):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        self.max_w, self.max_h, self.n_slots, self.max_pts, self.device = max_w, max_h, n_slots, max_pts, device
        check(self._lib.icelk_create(device, max_w, max_h, n_slots, max_pts, C.byref(self._h)), None)

    # -- lifetime -----------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.icelk_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _ck(self, rc):
        check(rc, self._h)

    def set_stream(self, stream_ptr):
        """Run on an existing HIP stream (e.g. torch.cuda.current_stream().cuda_stream)."""
        self._ck(self._lib.icelk_set_stream(self._h, C.c_void_p(stream_ptr or 0)))

    def sync(self):
        self._ck(self._lib.icelk_sync(self._h))

    def set_lk_kernel(self, which):
        """0 = default, LK_GENERIC_KERNEL or LK_MULTI_PER_WAVE: the three tracker kernels give identical results."""
        self._ck(self._lib.icelk_set_lk_kernel(self._h, int(which)))

    def set_variant(self, name, value):
        """A named build-dependent variant of OpenCV's arithmetic: "lk_sums" 0|1|2, "sobel_fma" 0..3, "eig_fma" 0|1
        (icelk_set_variant; 0 = default).  The oracle has the same switches (oracle.set_variant)."""
        self._ck(self._lib.icelk_set_variant(self._h, name.encode(), int(value)))

    def set_fb_distance(self, form):
        """FB_HYPOT (np.hypot on float32, s1:330; default) or FB_SQRT ((dx**2+dy**2)**0.5, s0_1:99)."""
        self._ck(self._lib.icelk_set_fb_distance(self._h, int(form)))

    # -- ingest -------------------------------------------------------------------------------
    def upload_gray(self, slot, img):
        a = _gray2d(img)
        self._ck(self._lib.icelk_upload_gray(self._h, slot, _u8(a), a.shape[1], a.shape[0], a.strides[0]))

    def upload_bgr(self, slot, img, variant=GRAY_CV4, crop=None):
        """3-channel frame -> gray in `slot` (s1:310-311).  `crop` = (left, top, right, bottom) pixels to drop, the
        box `Camera.crop_image` cuts (camtools.py:213-231): only the kept region crosses PCIe, straight out of the
        decoded frame (the reference's lossy JPEG re-save of the crop, s1:272, has no counterpart)."""
        a = np.asarray(img)
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
            raise ValueError("expected HxWx3 uint8 image")
        if crop is not None:
            left, top, right, bottom = (int(v) for v in crop)
            if min(left, top, right, bottom) < 0 or left + right >= a.shape[1] or top + bottom >= a.shape[0]:
                raise ValueError("crop box leaves no image")
            a = a[top:a.shape[0] - bottom, left:a.shape[1] - right]
        if a.strides[2] != 1 or a.strides[1] != 3 or a.strides[0] < 3 * a.shape[1]:
            a = np.ascontiguousarray(a)   # rows must be dense; a row pitch (cropped view) is fine as it is
        self._ck(self._lib.icelk_upload_bgr(self._h, slot, _u8(a), a.shape[1], a.shape[0], a.strides[0], variant))

    def set_gray_device(self, slot, dev_ptr, w, h, stride):
        self._ck(self._lib.icelk_set_gray_device(self._h, slot, C.c_void_p(dev_ptr), w, h, stride))

    def cvt_bgr_device(self, slot, dev_ptr, w, h, stride, variant=GRAY_CV4):
        self._ck(self._lib.icelk_cvt_bgr_device(self._h, slot, C.c_void_p(dev_ptr), w, h, stride, variant))

    def upload_gray_async(self, slot, pinned_ptr, w, h, stride):
        self._ck(self._lib.icelk_upload_gray_async(self._h, slot, C.c_void_p(pinned_ptr), w, h, stride))

    def host_alloc(self, nbytes):
        """Pinned host memory for `upload_gray_async` (address as int); release with `host_free`."""
        p = C.c_void_p()
        self._ck(self._lib.icelk_host_alloc(C.byref(p), int(nbytes)))
        return p.value

    def host_free(self, ptr):
        self._ck(self._lib.icelk_host_free(C.c_void_p(ptr)))

    def synth_frame(self, slot, w, h, ux=0, uy=0, seed=1234, affine=None):
        """Procedural frame on the device (bit-identical to synth.frame); `affine` = (ax, bx, ay, by) in 2^-20 px/px."""
        if affine is None:
            self._ck(self._lib.icelk_synth_frame(self._h, slot, w, h, int(ux), int(uy), int(seed)))
            return
        a = (C.c_int32 * 4)(*[int(v) for v in affine])
        self._ck(self._lib.icelk_synth_frame_affine(self._h, slot, w, h, int(ux), int(uy), int(seed), a))

    def drop_pyramid(self, slot):
        self._ck(self._lib.icelk_drop_pyramid(self._h, slot))

    def download_level(self, slot, level=0):
        w, h = C.c_int(0), C.c_int(0)
        self._ck(self._lib.icelk_download_level(self._h, slot, level, None, 0, C.byref(w), C.byref(h)))
        out = np.empty((h.value, w.value), np.uint8)
        self._ck(self._lib.icelk_download_level(self._h, slot, level, _u8(out), w.value, C.byref(w), C.byref(h)))
        return out

    def build_pyramid(self, slot, winSize=(21, 21), maxLevel=3):
        n = C.c_int(0)
        self._ck(self._lib.icelk_build_pyramid(self._h, slot, winSize[0], winSize[1], maxLevel, C.byref(n)))
        return n.value

    # -- tracker ------------------------------------------------------------------------------
    def pyrlk(self, prev_slot, next_slot, prev_pts, next_pts=None, winSize=(21, 21), maxLevel=3,
              criteria=DEFAULT_CRITERIA, flags=0, minEigThreshold=1e-4):
        p0 = np.ascontiguousarray(prev_pts, dtype=np.float32).reshape(-1, 2)
        n = p0.shape[0]
        if flags & OPTFLOW_USE_INITIAL_FLOW:
            if next_pts is None:
                raise ValueError("OPTFLOW_USE_INITIAL_FLOW needs nextPts")
            p1 = np.ascontiguousarray(next_pts, dtype=np.float32).reshape(-1, 2).copy()
            if p1.shape[0] != n:
                raise ValueError("nextPts and prevPts differ in length")
        else:
            p1 = np.zeros((n, 2), np.float32)
        st = np.zeros(n, np.uint8)
        er = np.zeros(n, np.float32)
        t, cnt, eps = _criteria(criteria)
        self._ck(self._lib.icelk_pyrlk(self._h, prev_slot, next_slot, _f32(p0), _f32(p1), _u8(st), _f32(er), n,
                                       winSize[0], winSize[1], maxLevel, t, cnt, eps, flags, minEigThreshold))
        return p1.reshape(-1, 1, 2), st.reshape(-1, 1), er.reshape(-1, 1)

    def track_fb(self, slot0, slot1, p0, winSize=(21, 21), maxLevel=3, criteria=DEFAULT_CRITERIA,
                 minEigThreshold=1e-4, fb_threshold=1.0):
        p0 = np.ascontiguousarray(p0, dtype=np.float32).reshape(-1, 2)
        n = p0.shape[0]
        out = dict(p1=np.zeros((n, 2), np.float32), p0r=np.zeros((n, 2), np.float32),
                   st_fwd=np.zeros(n, np.uint8), st_bwd=np.zeros(n, np.uint8),
                   err_fwd=np.zeros(n, np.float32), err_bwd=np.zeros(n, np.float32),
                   dist=np.zeros(n, np.float32), valid=np.zeros(n, np.uint8))
        t, cnt, eps = _criteria(criteria)
        self._ck(self._lib.icelk_track_fb(self._h, slot0, slot1, _f32(p0), n, winSize[0], winSize[1], maxLevel, t,
                                          cnt, eps, minEigThreshold, fb_threshold, _f32(out["p1"]),
                                          _f32(out["p0r"]), _u8(out["st_fwd"]), _u8(out["st_bwd"]),
                                          _f32(out["err_fwd"]), _f32(out["err_bwd"]), _f32(out["dist"]),
                                          _u8(out["valid"])))
        return out

    def fb_filter(self, p0, p0r, fb_threshold=1.0):
        """(dist, valid) of s1:329-333 for given p0 / p0r, computed by the tracker's own device function."""
        a = np.ascontiguousarray(p0, dtype=np.float32).reshape(-1, 2)
        b = np.ascontiguousarray(p0r, dtype=np.float32).reshape(-1, 2)
        if a.shape != b.shape:
            raise ValueError("p0 and p0r differ in length")
        n = a.shape[0]
        dist, valid = np.zeros(n, np.float32), np.zeros(n, np.uint8)
        self._ck(self._lib.icelk_fb_filter(self._h, _f32(a), _f32(b), n, float(fb_threshold), _f32(dist), _u8(valid)))
        return dist, valid

    # -- detector -----------------------------------------------------------------------------
    def set_mask(self, mask):
        if mask is None:
            self._ck(self._lib.icelk_set_mask(self._h, None, 0, 0, 0))
            return
        m = _gray2d(mask, "mask")
        self._ck(self._lib.icelk_set_mask(self._h, _u8(m), m.shape[1], m.shape[0], m.strides[0]))

    def set_mask_polygon(self, poly, crop_left, crop_top, w, h):
        """The mask of s1:285-291 rasterised on the device from `maskpoly` (camtools.py:184-211)."""
        p = np.ascontiguousarray(poly, dtype=np.float64).reshape(-1, 2)
        self._ck(self._lib.icelk_set_mask_polygon(self._h, p.ctypes.data_as(_lib.f64p), len(p), float(crop_left),
                                                  float(crop_top), int(w), int(h)))

    def download_mask(self):
        w, h = C.c_int(0), C.c_int(0)
        self._ck(self._lib.icelk_download_mask(self._h, None, 0, C.byref(w), C.byref(h)))
        m = np.empty((h.value, w.value), np.uint8)
        self._ck(self._lib.icelk_download_mask(self._h, _u8(m), w.value, C.byref(w), C.byref(h)))
        return m

    def min_eig_map(self, slot, blockSize=3):
        lvl = self.download_level(slot, 0)
        out = np.empty(lvl.shape, np.float32)
        self._ck(self._lib.icelk_min_eig_map(self._h, slot, blockSize, _f32(out), out.shape[1]))
        return out

    def good_features(self, slot, maxCorners, qualityLevel, minDistance, use_mask=False, blockSize=3):
        cap = self.max_pts if maxCorners <= 0 else min(int(maxCorners), self.max_pts)
        out = np.empty((max(cap, 1), 2), np.float32)
        n = C.c_int(0)
        self._ck(self._lib.icelk_good_features(self._h, slot, 1 if use_mask else 0, int(maxCorners),
                                               float(qualityLevel), float(minDistance), int(blockSize), _f32(out), cap,
                                               C.byref(n)))
        if n.value == 0:
            return None
        return out[:n.value].reshape(-1, 1, 2).copy()

    def detect_fast_stats(self, w, h):
        out = (C.c_longlong * 8)()
        self._ck(self._lib.icelk_detect_fast_stats(self._h, int(w), int(h), out))
        return dict(tiles=out[0], listed=out[1], whole_tiles=out[2], ties=out[3], max_listed=out[4], max_overflow_tiles=out[5],
                    longest_list=out[6], evaluated=out[7])

    def detect_stats(self):
        a, b = C.c_int(0), C.c_int(0)
        self._ck(self._lib.icelk_detect_stats(self._h, C.byref(a), C.byref(b)))
        return dict(candidates=a.value, accepted=b.value)

    # -- device-resident segment state ----------------------------------------------------------
    def seg_detect(self, slot, maxCorners, qualityLevel, minDistance, use_mask=False, blockSize=3):
        n = C.c_int(0)
        self._ck(self._lib.icelk_seg_detect(self._h, slot, 1 if use_mask else 0, int(maxCorners), float(qualityLevel),
                                            float(minDistance), int(blockSize), C.byref(n)))
        return n.value

    def build_pyramid_ahead(self, slot, winSize=(21, 21), maxLevel=3):
        self._ck(self._lib.icelk_build_pyramid_ahead(self._h, slot, winSize[0], winSize[1], maxLevel))

    def seg_detect_prepare(self, slot, use_mask=False, blockSize=3):
        self._ck(self._lib.icelk_seg_detect_prepare(self._h, slot, 1 if use_mask else 0, int(blockSize)))

    def seg_detect_begin(self, slot, maxCorners, qualityLevel, minDistance, use_mask=False, blockSize=3):
        self._ck(self._lib.icelk_seg_detect_begin(self._h, slot, 1 if use_mask else 0, int(maxCorners),
                                                  float(qualityLevel), float(minDistance), int(blockSize)))

    def seg_detect_finish(self, maxCorners):
        n = C.c_int(0)
        self._ck(self._lib.icelk_seg_detect_finish(self._h, int(maxCorners), C.byref(n)))
        return n.value

    def seg_detect_stage(self, maxCorners):
        n = C.c_int(0)
        self._ck(self._lib.icelk_seg_detect_stage(self._h, int(maxCorners), C.byref(n)))
        return n.value

    def seg_detect_stage_try(self, maxCorners):
        """icelk_seg_detect_stage_try: the corner count once the oldest detection in flight is through, else None (no wait)."""
        n, done = C.c_int(0), C.c_int(0)
        self._ck(self._lib.icelk_seg_detect_stage_try(self._h, int(maxCorners), C.byref(n), C.byref(done)))
        return n.value if done.value else None

    def seg_detect_cancel(self):
        """Abandon detections begun / prepared / staged ahead and never used (icelk_seg_detect_cancel)."""
        self._ck(self._lib.icelk_seg_detect_cancel(self._h))

    def seg_switch(self):
        self._ck(self._lib.icelk_seg_switch(self._h))

    def seg_track(self, slot_prev, slot_next, winSize=(21, 21), maxLevel=3, criteria=DEFAULT_CRITERIA,
                  minEigThreshold=1e-4, fb_threshold=1.0, wait=True):
        t, cnt, eps = _criteria(criteria)
        if wait:
            n = C.c_int(0)
            self._ck(self._lib.icelk_seg_track(self._h, slot_prev, slot_next, winSize[0], winSize[1], maxLevel, t, cnt,
                                               eps, minEigThreshold, fb_threshold, C.byref(n)))
            return n.value
        self._ck(self._lib.icelk_seg_track_async(self._h, slot_prev, slot_next, winSize[0], winSize[1], maxLevel, t,
                                                 cnt, eps, minEigThreshold, fb_threshold))
        return None

    def seg_track_defer(self, slot_prev, slot_next, winSize=(21, 21), maxLevel=3, criteria=DEFAULT_CRITERIA,
                        minEigThreshold=1e-4, fb_threshold=1.0):
        """The last pair of a segment: nothing is launched, the pair goes out together with the first pair of the next
        segment (one tracker launch for both) -- see icelk_seg_track_defer in include/icelk.h."""
        t, cnt, eps = _criteria(criteria)
        self._ck(self._lib.icelk_seg_track_defer(self._h, slot_prev, slot_next, winSize[0], winSize[1], maxLevel, t,
                                                 cnt, eps, minEigThreshold, fb_threshold))

    def seg_flush(self):
        self._ck(self._lib.icelk_seg_flush(self._h))

    def seg_template_stats(self):
        """(pairs whose forward pass took the templates of the pair before, pairs that left templates) -- diagnostics."""
        out = (C.c_longlong * 2)()
        self._ck(self._lib.icelk_seg_template_stats(self._h, out))
        return int(out[0]), int(out[1])

    def seg_template_info(self):
        """(bytes of one template table, tracks it has room for, state: 0 in use / 1 switched off / 2 allocation failed)."""
        by, rows, st = C.c_longlong(0), C.c_longlong(0), C.c_int(0)
        self._ck(self._lib.icelk_seg_template_info(self._h, C.byref(by), C.byref(rows), C.byref(st)))
        return int(by.value), int(rows.value), int(st.value)

    def seg_tail_stats(self):
        """(segments staged by the device-driven detection tail, segments staged by the host's tail) -- diagnostics."""
        out = (C.c_longlong * 2)()
        self._ck(self._lib.icelk_seg_tail_stats(self._h, out))
        return int(out[0]), int(out[1])

    def seg_track_len_hint(self, track_len):
        """Pairs per segment of the driving loop (0: unknown); lets the last pair of a segment skip leaving templates for a
        successor that never comes (icelk_seg_track_len_hint).  Results do not depend on it."""
        self._ck(self._lib.icelk_seg_track_len_hint(self._h, int(track_len)))

    def seg_live(self):
        n, tot = C.c_int(0), C.c_int64(0)
        self._ck(self._lib.icelk_seg_live(self._h, C.byref(n), C.byref(tot)))
        return n.value, tot.value

    def seg_archive(self, dev_tracks_ptr, dev_quality_ptr, dev_count_ptr, cap_rows, closed=False):
        """Gather the current segment's surviving tracks (`closed`: those of the segment the latest switch closed) into
        device memory of the caller (no wait); returns the vertex count the rows have."""
        nv = C.c_int(0)
        fn = self._lib.icelk_seg_archive_closed if closed else self._lib.icelk_seg_archive
        self._ck(fn(self._h, C.c_void_p(dev_tracks_ptr), C.c_void_p(dev_quality_ptr or 0),
                                             C.c_void_p(dev_count_ptr), int(cap_rows), C.byref(nv)))
        return nv.value

    def seg_read(self, closed=False):
        """(tracks (n, V, 2) f32, trackquality (n, V-1) f32): what np.savez stores at s1:394-395.  `closed`: of the
        segment the latest switch closed instead of the current one."""
        n, nv = C.c_int(0), C.c_int(0)
        fn = self._lib.icelk_seg_read_closed if closed else self._lib.icelk_seg_read
        self._ck(fn(self._h, None, None, 0, 0, C.byref(n), C.byref(nv)))
        tracks = np.zeros((n.value, nv.value, 2), np.float32)
        quality = np.zeros((n.value, max(nv.value - 1, 0)), np.float32)
        if n.value:
            self._ck(fn(self._h, _f32(tracks), _f32(quality), n.value, nv.value, C.byref(n),
                                              C.byref(nv)))
        return tracks, quality

    # -- measurement ----------------------------------------------------------------------------
    def prof_enable(self, on=True):
        """on: False / True (every kernel) / 2 (tracker launches only: the other streams carry no event records)."""
        self._ck(self._lib.icelk_prof_enable(self._h, int(on)))

    def prof_reset(self):
        self._ck(self._lib.icelk_prof_reset(self._h))

    def prof_iterations(self):
        """(forward, backward) LK iteration counts per feature of the latest tracker call made while profiling was on;
        tracks that were already dead are left out."""
        n = C.c_int(0)
        self._ck(self._lib.icelk_prof_iterations(self._h, None, 0, C.byref(n)))
        buf = np.zeros(max(n.value, 1), np.uint32)
        if n.value:
            self._ck(self._lib.icelk_prof_iterations(self._h, buf.ctypes.data_as(C.POINTER(C.c_uint32)), n.value,
                                                     C.byref(n)))
        buf = buf[:n.value]
        buf = buf[buf != 0xffffffff]
        return (buf & 0xffff).astype(np.int64), (buf >> 16).astype(np.int64)

    def stream_probe_info(self):
        """Which hardware queues the side streams of this handle landed on (icelk_stream_probe_info)."""
        picks = (C.c_int * 4)()
        q, lim = C.c_double(0), C.c_double(0)
        self._ck(self._lib.icelk_stream_probe_info(self._h, picks, C.byref(q), C.byref(lim)))
        return dict(detection=picks[0], candidates=picks[1], pyramid=picks[2], tail=picks[3], quickest=q.value,
                    limit=lim.value, probed=picks[0] >= 0)

    def prof_table(self):
        out = {}
        for k in range(self._lib.icelk_prof_count()):
            n, ms = C.c_int(0), C.c_double(0)
            self._ck(self._lib.icelk_prof_get(self._h, k, C.byref(n), C.byref(ms)))
            if n.value:
                out[self._lib.icelk_prof_name(k).decode()] = dict(launches=n.value, total_ms=ms.value,
                                                                  avg_us=1e3 * ms.value / n.value)
        return out
