"""One folder of time-lapse photographs -> track files: the body of `lucaskanade_tracking` in the reference
(s1_lucaskanade_tracking.py:234-450) with the frame loop on the GPU.

What the reference does per day folder: crop every photo with PIL and save it again as JPEG (s1:272,
camtools.py:64-104), list the cropped copies, build the fjord mask (s1:285-294), then for every `start` offset walk
the list: decode (s1:310), cvtColor (s1:311), track / filter / extend (s1:313-359), at every `track_len`-th frame
check the time gaps and save the segment (s1:362-395), detect new corners (s1:437-448).

Here: JPEG decode stays on the host (PIL, a small thread pool decoding ahead -- it is two orders of magnitude slower
than the GPU step and is the real end-to-end bound); the crop box is cut during the upload of the decoded frame
(`Context.upload_bgr(crop=...)`), so the reference's lossy re-save of the crop has no counterpart and pixel values
are those of the original photo; gray conversion, detection, tracking, filtering and the track table are the
device-resident loop of `SegmentTracker`; the mask is rasterised on the device from the polygon
(`icelk_set_mask_polygon`) or uploaded.  Output files carry the reference's names and arrays.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .tracker import REF_FEATURE_PARAMS, REF_LK_PARAMS, SegmentTracker, npz_name, save_tracks, segment_time_ok


def _decode(path):
    from PIL import Image
    return np.array(Image.open(path))                     # s1:310 (RGB order; cvtColor is asked for BGR2GRAY)


def track_image_sequence(imagelist, target_dir, track_len, track_len_sec, startlist=(0,), crop=None, mask=None,
                         mask_polygon=None, feature_params=None, lk_params=None, decode_threads=4, decode_ahead=6,
                         gray_variant=4, device=0, on_segment=None, save=True):
    """Track one day's photos.  Returns [(npz path, tracks (n, T+1, 2) f32, trackquality (n, T) f32)] of the
    segments that pass the time-gap rule, in order.

    imagelist      sorted photo paths named '%Y%m%d-%H%M%S.jpg' (s1:264)
    crop           (left, top, right, bottom) of the calibration workbook (camtools.py:147-150) or None
    mask           H x W uint8 array (255 = detect here), or
    mask_polygon   (maskpoly, cropleft, croptop): rasterised on the device as s1:285-291 / camtools.py:184-211 do
    startlist      offsets into the list, each walked separately (s1:304)
    on_segment     optional callback(npz path, tracks, trackquality), e.g. a projection step
    """
    imagelist = [str(p) for p in imagelist]
    out = []
    if len(imagelist) <= track_len:                       # s1:267
        return out
    fp = dict(REF_FEATURE_PARAMS if feature_params is None else feature_params)
    lk = dict(REF_LK_PARAMS if lk_params is None else lk_params)
    first = _decode(imagelist[0])
    h, w = first.shape[0], first.shape[1]
    if crop is not None:
        left, top, right, bottom = (int(v) for v in crop)
        w, h = w - left - right, h - top - bottom
    trk = None
    try:
        with ThreadPoolExecutor(max_workers=max(1, int(decode_threads))) as pool:
            for start in startlist:
                names = imagelist[start:]
                if trk is not None:
                    trk.close()
                # a fresh loop state per start offset (the reference carries the last frame of the previous pass
                # into the first step of the next one; nothing is saved from that pair, s1:362-363)
                trk = SegmentTracker(w, h, track_len, feature_params=fp, lk_params=lk, mask=mask,
                                     mask_polygon=mask_polygon, device=device)
                pending = [pool.submit(_decode, p) for p in names[:decode_ahead]]
                for counter in range(len(names)):
                    frame = pending.pop(0).result()
                    if counter + decode_ahead < len(names):
                        pending.append(pool.submit(_decode, names[counter + decode_ahead]))
                    seg = trk.push_bgr(frame, variant=gray_variant, crop=crop)
                    if seg is None:
                        continue
                    seg_first, tracks, quality = seg
                    seg_names = names[seg_first:seg_first + track_len + 1]
                    if not segment_time_ok(seg_names, track_len_sec):        # s1:364-390
                        continue
                    npz = os.path.join(target_dir, npz_name(os.path.basename(seg_names[0]), track_len, track_len_sec))
                    if save:
                        save_tracks(npz, tracks, quality)
                    if on_segment is not None:
                        on_segment(npz, tracks, quality)
                    out.append((npz, tracks, quality))
    finally:
        if trk is not None:
            trk.close()
    return out
