"""MI355X-native sparse Lucas-Kanade tracking: the hot path of glacierbliss/iceberg_tracking_code
(s1_lucaskanade_tracking.py:307-450) as hand-written HIP kernels behind a C ABI (include/icelk.h).

    import iceberg_tracking_code_amd as cv2          # cvtColor / goodFeaturesToTrack / calcOpticalFlowPyrLK
    from iceberg_tracking_code_amd import SegmentTracker   # device-resident form of the same loop

There is no CPU fallback: calls raise if libicelk.so has not been built or no GPU is present.
"""
from .api import (COLOR_BGR2GRAY, COLOR_RGB2GRAY, calcOpticalFlowPyrLK, cvtColor, default_context,  # noqa: F401
                  goodFeaturesToTrack, release, set_device, set_gray_variant, set_variant)
from .context import (Context, DEFAULT_CRITERIA, GRAY_CV3, GRAY_CV4, OPTFLOW_LK_GET_MIN_EIGENVALS,  # noqa: F401
                      OPTFLOW_USE_INITIAL_FLOW, TERM_CRITERIA_COUNT, TERM_CRITERIA_EPS, TERM_CRITERIA_MAX_ITER)
from .tracker import (REF_FB_THRESHOLD, REF_FEATURE_PARAMS, REF_LK_PARAMS, SegmentTracker,  # noqa: F401
                      npz_name, save_tracks, segment_time_ok)
from .utm import CameraModel, REF_UTM_FILTER, cam_to_utm, project_segment, project_tracks, utm_name  # noqa: F401
from .sequence import track_image_sequence  # noqa: F401
from .gridding import bin_velocities, create_grid_across_fjord, points_in_polygon  # noqa: F401
from ._lib import IcelkError  # noqa: F401

__version__ = "0.1.0"
