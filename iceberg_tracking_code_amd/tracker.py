"""Host-side mirror of the reference's tracking loops, driving the HIP library.

`SegmentTracker` is the loop of s1_lucaskanade_tracking.py:304-450 (s0_1_test_lucaskanade_tracking.py:77-181) with
the state resident on the GPU (slots, pyramids, live points, track table): one pyramid build per frame instead of
the four OpenCV does, one fused forward+backward launch per pair, nothing crossing PCIe except the finished
segment.  The list-of-lists form of the same loop, written against cv2-shaped functions, is test harness and
lives in tests/reference_loops.py.

Also here: the `.npz` wire format consumed by s2_cam_to_utm.py:177,197-198,233-234.
"""
import datetime as dt
import os

import numpy as np

from .context import Context, TERM_CRITERIA_COUNT, TERM_CRITERIA_EPS

# parameter literals of the reference (s1:240-248, s0_1:37-45)
REF_FEATURE_PARAMS = dict(maxCorners=50000000, qualityLevel=0.007, minDistance=10, blockSize=10)
REF_LK_PARAMS = dict(winSize=(35, 35), maxLevel=4, criteria=(TERM_CRITERIA_EPS | TERM_CRITERIA_COUNT, 25, 0.03))
REF_FB_THRESHOLD = 1.0   # `valid = dist < 1` (s1:333), `self.distthreshold = 1.0` (s0_1:51)


def npz_name(first_image_path, track_len, track_len_sec):
    """'<%Y%m%d-%H%M%S>_{T*dt}sec_at_{dt}sec_tracks.npz' next to the image (s1:394)."""
    return "{}_{}sec_at_{}sec_tracks.npz".format(str(first_image_path).split(".")[0], track_len * track_len_sec,
                                                 track_len_sec)


def segment_time_ok(names, track_len_sec):
    """The +-2 s gap rule of s1:364-388 over the file names of one segment's frames."""
    times = [dt.datetime.strptime(os.path.basename(str(n)), "%Y%m%d-%H%M%S.jpg") for n in names]
    for a, b in zip(times[:-1], times[1:]):
        if (b - a).seconds not in (track_len_sec - 2, track_len_sec - 1, track_len_sec, track_len_sec + 1,
                                   track_len_sec + 2):
            return False
    return True


def save_tracks(npzname, tracks, trackquality):
    """np.savez(npzname, tracks=tracks, trackquality=trackquality) (s1:395)."""
    np.savez(npzname, tracks=tracks, trackquality=trackquality)


class SegmentTracker:
    """Device-resident form of the s1 loop.  Feed frames one at a time; finished segments come back as
    (first_frame_index, tracks (n, T+1, 2) f32, trackquality (n, T) f32) -- the np.savez payload.

    Frames are given as host arrays (`push`), as device pointers (`push_device`) or generated on the
    device (`push_synth`).  Three slots rotate: previous frame, current frame, and one being uploaded.
    """

    def __init__(self, width, height, track_len, feature_params=None, lk_params=None, mask=None, max_pts=1 << 18,
                 device=0, fb_threshold=REF_FB_THRESHOLD, ctx=None, n_slots=3, lookahead=True, mask_polygon=None,
                 pair_launch=True):
        self.track_len = int(track_len)
        if self.track_len < 1 or self.track_len > 16:
            raise ValueError("track_len must be in 1..16")
        self.fp = dict(REF_FEATURE_PARAMS if feature_params is None else feature_params)
        self.lk = dict(REF_LK_PARAMS if lk_params is None else lk_params)
        self.fb_threshold = float(fb_threshold)
        self.w, self.h = width, height
        self.ctx = ctx if ctx is not None else Context(width, height, n_slots=n_slots, max_pts=max_pts, device=device)
        self.n_slots = self.ctx.n_slots
        self.ctx.seg_track_len_hint(track_len)    # the last pair of a segment leaves no templates behind
        self.use_mask = mask is not None or mask_polygon is not None
        if mask_polygon is not None:
            # (maskpoly, cropleft, croptop): the mask of s1:285-291 rasterised on the device (camtools.py:184-211)
            poly, crop_left, crop_top = mask_polygon
            self.ctx.set_mask_polygon(poly, crop_left, crop_top, width, height)
        elif mask is not None:
            self.ctx.set_mask(mask)
        self.counter = 0          # frames consumed
        self.cur = -1             # slot of the newest frame
        self.seg_first = 0
        self.active = False
        self.n_detected = 0
        self._prefetched = []     # slots holding frames whose upload was started ahead of time
        self.lookahead = bool(lookahead)
        self.pair_launch = bool(pair_launch) and self.lookahead   # see `_step`: joint launch across a segment change
        self._pyr_ahead = set()   # slots whose pyramid was enqueued ahead of their step
        # how many steps ahead of a detection frame its min-distance stage / its corner candidates may start (`_step`)
        self.begin_ahead, self.prepare_ahead, self.stage_lag = 4, 6, 2
        # frames ahead whose pyramids are enqueued (2: what the next joint launch needs; 3: one more, so that the pyramid of
        # the frame after the next detection frame is not enqueued by the step right before the launch that needs it)
        self.pyramids_ahead = int(os.environ.get("ICELK_PYRAMIDS_AHEAD", "2"))
        self.stage_nowait = True  # the host round trip of a detection is taken without waiting (icelk_seg_detect_stage_try)
        # the step (counted back from the detection frame) at whose end the host WAITS for a detection's counts if they have
        # not come by themselves.  0: never before the frame itself -- the tail of a detection runs on the device without the
        # host (k_tail.hip), so adopting it is a look at pinned memory, done right before the switch.  1: at the end of step
        # d-1, as long as the tail needed the host (ICELK_HOST_TAIL=1: it had to be enqueued a step before it was needed)
        self.stage_block_at = 1 if os.environ.get("ICELK_HOST_TAIL") else 0
        if os.environ.get("ICELK_STAGE_BLOCK_AT"):     # A/B measurements
            self.stage_block_at = int(os.environ["ICELK_STAGE_BLOCK_AT"])
        self._resident = False    # inside push_slot
        # callable(first_frame, closed) invoked once per finished segment, when all its pairs have been launched: e.g.
        # ctx.seg_archive(..., closed=closed).  closed=False: the segment is still the current one (the switch follows);
        # closed=True: the switch has happened (its last pair went out in a joint launch, see `_step`)
        self.on_close = None
        self._advanced = False    # the pair (cur, next) has gone out already, with the joint launch of this step
        self._advanced_slot = None   # ... and `next` was announced to sit in this slot
        self._det_queue = []      # detections in flight, oldest first: (frame counter, step at which it was begun)
        self._begun_upto = -1     # latest frame whose detection has been begun
        self._prep_upto = -1      # latest frame whose corner candidates have been prepared ahead
        self._staged = False      # a new segment waits in the spare set for the switch
        self._staged_for = None   # ... the detection frame it belongs to
        self._staged_n = 0
        self.pairs_launched = 0   # frame pairs handed to the tracker kernels so far (a joint launch carries two)

    # -- frame sources --------------------------------------------------------------------------
    def _next_slot(self):
        if self._prefetched:
            raise RuntimeError("prefetched frames are pending; consume them with push_prefetched()")
        return (self.cur + 1) % self.n_slots

    def push(self, frame_gray, wait=True):
        s = self._next_slot()
        self.ctx.upload_gray(s, frame_gray)
        return self._step(s, wait)

    def push_bgr(self, frame, wait=True, variant=4, crop=None):
        """`crop` = (left, top, right, bottom): the box of camtools.py:213-231, cut during the upload."""
        s = self._next_slot()
        self.ctx.upload_bgr(s, frame, variant, crop)
        return self._step(s, wait)

    def push_device(self, dev_ptr, stride, wait=True):
        s = self._next_slot()
        self.ctx.set_gray_device(s, dev_ptr, self.w, self.h, stride)
        return self._step(s, wait)

    def push_pinned(self, pinned_ptr, stride, wait=True):
        s = self._next_slot()
        self.ctx.upload_gray_async(s, pinned_ptr, self.w, self.h, stride)
        return self._step(s, wait)

    def prefetch_pinned(self, pinned_ptr, stride):
        """Start the upload of a FUTURE frame from pinned host memory (BASELINE.json configs[2]: hipMemcpyAsync
        double buffering).  Frames are consumed in prefetch order by `push_prefetched`; with n_slots slots at
        most n_slots - 2 uploads may be in flight (previous and current frame stay resident)."""
        if len(self._prefetched) >= self.n_slots - 2:
            raise RuntimeError("too many frames in flight for %d slots" % self.n_slots)
        last = self._prefetched[-1] if self._prefetched else self.cur
        s = (last + 1) % self.n_slots
        self.ctx.upload_gray_async(s, pinned_ptr, self.w, self.h, stride)
        self._prefetched.append(s)

    def push_prefetched(self, wait=True):
        if not self._prefetched:
            raise RuntimeError("no prefetched frame")
        s = self._prefetched.pop(0)
        q = self._prefetched
        return self._step(s, wait, *q[:self.MAX_AHEAD])

    def push_slot(self, slot, wait=True, *ahead):
        """Use a frame that already sits in `slot` (level 0 resident in HBM); its pyramid is rebuilt.
        `ahead`: the slots where the following frames already sit, nearest first, up to MAX_AHEAD of them (see `_step`);
        None ends the list."""
        if slot not in self._pyr_ahead:
            self.ctx.drop_pyramid(slot)
        self._resident = True
        try:
            return self._step(slot, wait, *ahead)
        finally:
            self._resident = False

    def push_synth(self, ux, uy, seed=1234, wait=True):
        s = self._next_slot()
        self.ctx.synth_frame(s, self.w, self.h, ux, uy, seed)
        return self._step(s, wait)

    # -- the loop body (s1:313-450) -------------------------------------------------------------
    def _detect_begin(self, slot):
        self.ctx.seg_detect_begin(slot, self.fp["maxCorners"], self.fp["qualityLevel"], self.fp["minDistance"],
                                  self.use_mask, self.fp.get("blockSize", 3))

    MAX_AHEAD = 6

    def _step(self, slot, wait, *ahead_slots):
        """One pass of the loop body.  `ahead_slots`: slots of the FOLLOWING frames (nearest first, up to six) when they
        are already on their way to the device (prefetched uploads, resident ring).  The detector needs nothing but its
        own frame, so the work for a coming detection frame d is spread over the steps before it and runs beside their
        tracker launches, each part as early as the frame's slot is known and the buffers it needs are free:
          d-6  corner candidates (`seg_detect_prepare`: a spare candidate buffer, own stream)
          d-4  min-distance stage (`seg_detect_begin`) -- two detections may be in flight
          d-2  the one host round trip of a detection, the sort and the new segment's initialisation in the spare set
               of segment buffers (`seg_detect_stage`): it waits for kernels that were issued two steps -- with
               track_len 2 a whole tracker launch -- earlier, and the min-distance stage of the NEXT detection is on
               the device already, so the host never stands in a loop with the detector's kernels
          d    the switch (no GPU work, no wait).
        With less lookahead the same calls move later (a stage one step after its begin at the least); with none,
        everything happens at d.

        Joint launch: the last pair of the closing segment, (d-1, d), and the first pair of the new one, (d, d+1), are
        independent (s1:362 / s1:440).  When frame d+1 is already on the device and nothing has to be read at d, step d
        holds the first back (`seg_track_defer`), switches, and tracks (d, d+1) at once: both pairs go to the device as
        ONE tracker launch and step d+1 has no pair left to launch.  Results are those of the serial order in every
        case."""
        out = None
        prev = self.cur
        T = self.track_len
        c = self.counter
        detect = c % T == 0
        ahead = self.lookahead and T >= 2
        bs = self.fp.get("blockSize", 3)
        lk_tail = (self.lk["winSize"], self.lk["maxLevel"], self.lk["criteria"], self.lk.get("minEigThreshold", 1e-4),
                   self.fb_threshold)
        slot_of = {0: slot}
        for k, s_k in enumerate(ahead_slots[:self.MAX_AHEAD]):
            if s_k is None:
                break
            slot_of[k + 1] = s_k
        next_slot, next2_slot = slot_of.get(1), slot_of.get(2)
        self._pyr_ahead.discard(slot)
        staged_now = self._staged and self._staged_for == c
        if detect and ahead and not staged_now and not self._staged and self._det_queue and self._det_queue[0][0] == c \
                and self._det_queue[0][1] < c:
            # the detection of this frame was begun steps ago and its tail has run on the device: adopt it now (waits only if
            # the device is behind)
            self._staged_n = self.ctx.seg_detect_stage(self.fp["maxCorners"])
            self._staged, self._staged_for = True, c
            self._det_queue.pop(0)
            staged_now = True
        if detect and not staged_now and not (self._det_queue and self._det_queue[0][0] == c):
            # nothing was started ahead for this detection frame: start it now, on its own stream, so that it runs
            # beside the tracker launch below (the reference does them back to back, s1:323-326 then s1:437)
            self._detect_begin(slot)
            self._det_queue.append((c, c))
            self._begun_upto = c
        joint = False
        if self._advanced:
            # the pair (prev, slot) went out with the launch of the previous step -- on the strength of the slot that was
            # announced for this frame then
            if slot != self._advanced_slot:
                raise RuntimeError("frame pushed in slot %d, but slot %d was announced for it one step ago (its pair has "
                                   "been launched already)" % (slot, self._advanced_slot))
            self._advanced = False
        elif self.active:
            joint = detect and self.pair_launch and not wait and staged_now and next_slot is not None and c > 0
            if joint:
                self.ctx.seg_track_defer(prev, slot, *lk_tail)
                self.ctx.seg_switch()
                self.ctx.seg_track(slot, next_slot, *lk_tail, wait=False)
                self.pairs_launched += 2
                self._advanced, self._advanced_slot = True, next_slot
                if self.on_close is not None:
                    self.on_close(self.seg_first, True)
                self.n_detected = self._staged_n
                self._staged = False
                self.seg_first = c
            else:
                self.ctx.seg_track(prev, slot, *lk_tail, wait=False)
                self.pairs_launched += 1
        if self.lookahead:
            # pyramids of the following frames, on the copy stream, in the shadow of the tracker launch above (two ahead
            # when a joint launch may need frame c+2 at step c+1)
            ahead_pyr = (next_slot, next2_slot if self.pair_launch else None)
            if self.pair_launch and self.pyramids_ahead >= 3:
                ahead_pyr += (slot_of.get(3),)
            for s_k in ahead_pyr:
                if s_k is None or s_k in self._pyr_ahead or s_k in (slot, prev) or (s_k == next_slot and self._advanced):
                    continue
                if self._resident:
                    self.ctx.drop_pyramid(s_k)   # a resident ring is rebuilt on every visit
                self.ctx.build_pyramid_ahead(s_k, self.lk["winSize"], self.lk["maxLevel"])
                self._pyr_ahead.add(s_k)
        if detect and not joint:
            if c > 0 and self.on_close is not None:
                self.on_close(self.seg_first, False)
            if c > 0 and wait:
                tracks, quality = self.ctx.seg_read()
                out = (self.seg_first, tracks, quality)
            if staged_now:
                self.ctx.seg_switch()
                self.n_detected = self._staged_n
                self._staged = False
            else:
                self.n_detected = self.ctx.seg_detect_finish(self.fp["maxCorners"])   # the oldest in flight: frame c's
                self._det_queue.pop(0)
            self.active = True
            self.seg_first = c
        if ahead:
            # the oldest detection in flight: its host round trip, and the new segment into the spare set (behind this
            # step's switch, if there was one: the spare set is the one after the current) -- once its kernels have had
            # `stage_lag` steps, or when its frame is next
            if self._det_queue and not self._staged:
                d, begun = self._det_queue[0]
                if d > c and begun < c and (c - begun >= self.stage_lag or d - c <= 1):
                    # without waiting while there is a later step to do it at; the segment is needed at step d
                    # (waiting only at d itself was measured too: the tail of the detection -- sort, corner list, the new
                    # segment's tables -- then starts when the tracker launch already needs it: C3 3 820 -> 3 290 pairs/s)
                    if d - c <= self.stage_block_at or not self.stage_nowait:
                        n_new = self.ctx.seg_detect_stage(self.fp["maxCorners"])
                    else:
                        n_new = self.ctx.seg_detect_stage_try(self.fp["maxCorners"])
                    if n_new is not None:
                        self._staged_n = n_new
                        self._staged, self._staged_for = True, d
                        self._det_queue.pop(0)
            # min-distance stage of the next detection frame not begun yet: up to `begin_ahead` steps ahead, two
            # detections in flight at most
            d = max(self._begun_upto, c) // T * T + T
            if len(self._det_queue) < 2 and 1 <= d - c <= self.begin_ahead and slot_of.get(d - c) is not None:
                self._detect_begin(slot_of[d - c])
                self._det_queue.append((d, c))
                self._begun_upto = d
            # corner candidates of the detection frame after the one begun last, once the frame's slot is known (six steps
            # ahead at most) and the candidates prepared before have been adopted by their seg_detect_begin
            d = max(self._begun_upto, self._prep_upto, c) // T * T + T
            if self._prep_upto <= self._begun_upto and 1 <= d - c <= self.prepare_ahead and slot_of.get(d - c) is not None:
                self.ctx.seg_detect_prepare(slot_of[d - c], self.use_mask, bs)
                self._prep_upto = d
        self.cur = slot
        self.counter += 1
        return out

    def flush(self):
        """Nothing of a pushed frame is held back across steps any more; kept so that callers can mark the end of a
        sequence (a pair waiting inside the library -- icelk_seg_track_defer used directly -- goes out)."""
        self.ctx.seg_flush()

    def live(self):
        return self.ctx.seg_live()

    def abort(self):
        """End (or interrupt) a sequence whose announced frames will not all be pushed: a pair held back inside the library
        goes out, detections begun / prepared / staged ahead for frames that never came are abandoned
        (icelk_seg_detect_cancel).  The current segment stays readable; the handle's one-call detector forms work again,
        and pushing may continue -- the next detection frame then starts its detection when it arrives.  Returns the
        number of frames consumed so far."""
        self.ctx.seg_flush()
        self.ctx.seg_detect_cancel()
        if self._advanced:
            # a joint launch has already tracked the pair into the frame announced for the next step (it sat in its slot):
            # that frame counts as consumed -- it is never a detection frame (track_len >= 2 wherever pairs are joined)
            self.cur, self.counter, self._advanced = self._advanced_slot, self.counter + 1, False
        self._det_queue = []
        self._staged, self._staged_for, self._staged_n = False, None, 0
        self._begun_upto = self._prep_upto = self.counter - 1
        self._pyr_ahead = set()
        return self.counter

    def close(self):
        self.ctx.close()
