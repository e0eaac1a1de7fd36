"""ctypes binding of libicelk.so -- the thin shim between the Python host code and the HIP kernels.

Signatures mirror include/icelk.h one to one.  There is no CPU fallback: if the library is missing
or no GPU is present the calls raise.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ICELK_LIBRARY: another build of this same library (A/B measurements of kernel variants: tools/build_variant.sh)
LIB_PATH = os.environ.get("ICELK_LIBRARY") or os.path.join(_HERE, "libicelk.so")

OK, EARG, ENOMEM, EHIP, ECAP, ESTATE = 0, -1, -2, -3, -4, -5

u8p = C.POINTER(C.c_uint8)
f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int)
i64p = C.POINTER(C.c_int64)
f64p = C.POINTER(C.c_double)
vp = C.c_void_p
handle_p = C.c_void_p

# name -> (restype, argtypes); the single source of truth for the symbol-export test as well
SIGNATURES = {
    "icelk_version": (C.c_int, []),
    "icelk_last_error": (C.c_char_p, [handle_p]),
    "icelk_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(handle_p)]),
    "icelk_destroy": (C.c_int, [handle_p]),
    "icelk_set_stream": (C.c_int, [handle_p, vp]),
    "icelk_sync": (C.c_int, [handle_p]),
    "icelk_set_fb_distance": (C.c_int, [handle_p, C.c_int]),
    "icelk_set_lk_kernel": (C.c_int, [handle_p, C.c_int]),
    "icelk_upload_gray": (C.c_int, [handle_p, C.c_int, u8p, C.c_int, C.c_int, C.c_int]),
    "icelk_upload_bgr": (C.c_int, [handle_p, C.c_int, u8p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "icelk_set_gray_device": (C.c_int, [handle_p, C.c_int, vp, C.c_int, C.c_int, C.c_int]),
    "icelk_cvt_bgr_device": (C.c_int, [handle_p, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "icelk_upload_gray_async": (C.c_int, [handle_p, C.c_int, vp, C.c_int, C.c_int, C.c_int]),
    "icelk_host_alloc": (C.c_int, [C.POINTER(vp), C.c_uint64]),
    "icelk_host_free": (C.c_int, [vp]),
    "icelk_synth_frame": (C.c_int, [handle_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_uint32]),
    "icelk_synth_frame_affine": (C.c_int, [handle_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_uint32, i32p]),
    "icelk_drop_pyramid": (C.c_int, [handle_p, C.c_int]),
    "icelk_download_level": (C.c_int, [handle_p, C.c_int, C.c_int, u8p, C.c_int, i32p, i32p]),
    "icelk_build_pyramid": (C.c_int, [handle_p, C.c_int, C.c_int, C.c_int, C.c_int, i32p]),
    "icelk_pyrlk": (C.c_int, [handle_p, C.c_int, C.c_int, f32p, f32p, u8p, f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                              C.c_int, C.c_int, C.c_double, C.c_int, C.c_double]),
    "icelk_track_fb": (C.c_int, [handle_p, C.c_int, C.c_int, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_double, C.c_double, C.c_float, f32p, f32p, u8p, u8p, f32p, f32p, f32p,
                                 u8p]),
    "icelk_fb_filter": (C.c_int, [handle_p, f32p, f32p, C.c_int, C.c_float, f32p, u8p]),
    "icelk_set_mask": (C.c_int, [handle_p, u8p, C.c_int, C.c_int, C.c_int]),
    "icelk_min_eig_map": (C.c_int, [handle_p, C.c_int, C.c_int, f32p, C.c_int]),
    "icelk_good_features": (C.c_int, [handle_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, f32p,
                                      C.c_int, i32p]),
    "icelk_detect_stats": (C.c_int, [handle_p, i32p, i32p]),
    "icelk_seg_detect": (C.c_int, [handle_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, i32p]),
    "icelk_set_mask_polygon": (C.c_int, [handle_p, f64p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int]),
    "icelk_download_mask": (C.c_int, [handle_p, u8p, C.c_int, i32p, i32p]),
    "icelk_build_pyramid_ahead": (C.c_int, [handle_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "icelk_seg_detect_prepare": (C.c_int, [handle_p, C.c_int, C.c_int, C.c_int]),
    "icelk_seg_detect_begin": (C.c_int, [handle_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int]),
    "icelk_seg_detect_finish": (C.c_int, [handle_p, C.c_int, i32p]),
    "icelk_seg_detect_cancel": (C.c_int, [handle_p]),
    "icelk_set_variant": (C.c_int, [handle_p, C.c_char_p, C.c_int]),
    "icelk_seg_detect_stage_try": (C.c_int, [handle_p, C.c_int, i32p, i32p]),
    "icelk_detect_fast_stats": (C.c_int, [handle_p, C.c_int, C.c_int, C.POINTER(C.c_longlong)]),
    "icelk_seg_detect_stage": (C.c_int, [handle_p, C.c_int, i32p]),
    "icelk_seg_switch": (C.c_int, [handle_p]),
    "icelk_seg_track": (C.c_int, [handle_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_double, C.c_double, C.c_float, i32p]),
    "icelk_seg_track_async": (C.c_int, [handle_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_double, C.c_double, C.c_float]),
    "icelk_seg_track_defer": (C.c_int, [handle_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_double, C.c_double, C.c_float]),
    "icelk_seg_flush": (C.c_int, [handle_p]),
    "icelk_seg_track_len_hint": (C.c_int, [handle_p, C.c_int]),
    "icelk_seg_template_stats": (C.c_int, [handle_p, C.POINTER(C.c_longlong)]),
    "icelk_seg_tail_stats": (C.c_int, [handle_p, C.POINTER(C.c_longlong)]),
    "icelk_seg_template_info": (C.c_int, [handle_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), i32p]),
    "icelk_seg_read_closed": (C.c_int, [handle_p, f32p, f32p, C.c_int, C.c_int, i32p, i32p]),
    "icelk_seg_archive_closed": (C.c_int, [handle_p, vp, vp, vp, C.c_int, i32p]),
    "icelk_seg_live": (C.c_int, [handle_p, i32p, i64p]),
    "icelk_seg_archive": (C.c_int, [handle_p, vp, vp, vp, C.c_int, i32p]),
    "icelk_project_tracks": (C.c_int, [handle_p, f32p, C.c_int, C.c_int, vp, vp, f64p, f64p, f64p, f64p, f64p, u8p]),
    "icelk_seg_project": (C.c_int, [handle_p, vp, vp, C.c_int, C.c_int, f64p, f64p, f64p, f64p, f64p, u8p, i32p, i32p]),
    "icelk_points_in_polygon": (C.c_int, [handle_p, f64p, C.c_int, f64p, C.c_int, u8p]),
    "icelk_grid_bin": (C.c_int, [handle_p, f64p, f64p, f64p, f64p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int,
                                 C.c_int, u8p, i32p, f64p, f64p, f64p]),
    "icelk_seg_read": (C.c_int, [handle_p, f32p, f32p, C.c_int, C.c_int, i32p, i32p]),
    "icelk_prof_enable": (C.c_int, [handle_p, C.c_int]),
    "icelk_prof_reset": (C.c_int, [handle_p]),
    "icelk_prof_iterations": (C.c_int, [handle_p, C.POINTER(C.c_uint32), C.c_int, i32p]),
    "icelk_stream_probe_info": (C.c_int, [handle_p, i32p, f64p, f64p]),
    "icelk_prof_count": (C.c_int, []),
    "icelk_prof_name": (C.c_char_p, [C.c_int]),
    "icelk_prof_get": (C.c_int, [handle_p, C.c_int, i32p, f64p]),
}

_lib = None


class IcelkError(RuntimeError):
    """Raised for ICELK_EHIP / ICELK_ECAP / ICELK_ESTATE; bad arguments raise ValueError."""


def load():
    """Load libicelk.so.  Raises if it has not been built -- there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IcelkError(
                "libicelk.so is missing (%s). Build it with `python -m iceberg_tracking_code_amd.build` "
                "(needs hipcc); this package has no CPU fallback." % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, handle=None):
    if rc == OK:
        return
    msg = load().icelk_last_error(handle)
    msg = msg.decode() if msg else ""
    text = "icelk error %d: %s" % (rc, msg)
    if rc == EARG:
        raise ValueError(text)
    if rc == ENOMEM:
        raise MemoryError(text)
    raise IcelkError(text)
