/*
 * icelk.h -- C ABI of the MI355X-native sparse Lucas-Kanade tracking library (libicelk.so).
 *
 * This is the drop-in boundary for the ONE hot path of glacierbliss/iceberg_tracking_code: the
 * per-frame loop of s1_lucaskanade_tracking.py:307-450 (twin: s0_1_test_lucaskanade_tracking.py:77-181).
 * The reference has no FFI layer; the seam is three cv2 calls.  Each entry point below names the
 * reference call site it replaces.  Plain C: pointers and sizes only, no C++/torch types, no exceptions.
 *
 * Conventions
 *   - every function returns ICELK_OK (0) or a negative ICELK_E* code; icelk_last_error() gives text.
 *   - the caller owns every host buffer; the library owns all device memory (frames, pyramids, points).
 *   - a handle is bound to one GPU and is not thread-safe; use one handle per thread/process.
 *     Multi-GPU = one process per GPU, one handle each (DESIGN.md "Multi-GPU").
 *   - a handle owns SEVEN HIP streams: compute (tracker launches, segment bookkeeping; replaceable by the
 *     caller's stream, icelk_set_stream), two copy streams (icelk_upload_gray_async alternates between them), pyramid
 *     (icelk_build_pyramid_ahead), detection (min-distance stage), tail (what follows a detection's host round trip:
 *     sort, corner list, the new segment's tables) and candidates (the corner kernel of icelk_seg_detect_prepare).
 *     They are ordered against each other by events inside the library; a call whose outputs are host buffers has
 *     finished with them when it returns.  icelk_sync waits for ALL of them (and launches a pair held back by
 *     icelk_seg_track_defer first): after it nothing of the handle reads or writes any slot, mask or point buffer.
 *   - "slot" = a device-resident frame with its Gaussian pyramid.  Slots let the caller keep the
 *     previous frame (prev_gray = frame_gray, s1:450) and its pyramid on the GPU instead of
 *     rebuilding both pyramids in every cv2.calcOpticalFlowPyrLK call as OpenCV does.
 *   - points are interleaved (x, y) float32, i.e. numpy (N,1,2) float32 as cv2 returns them.
 *   - calls are synchronous with respect to their host outputs unless named *_async.
 */
#ifndef ICELK_H
#define ICELK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct icelk_ctx icelk_t;

#define ICELK_OK 0
#define ICELK_EARG (-1)    /* bad argument                       -> Python ValueError   */
#define ICELK_ENOMEM (-2)  /* host or device allocation failed   -> Python MemoryError  */
#define ICELK_EHIP (-3)    /* HIP runtime error                  -> Python RuntimeError */
#define ICELK_ECAP (-4)    /* exceeds the capacity given at icelk_create                */
#define ICELK_ESTATE (-5)  /* slot empty / pyramid missing / segment not started        */

/* cv2.TERM_CRITERIA_COUNT / cv2.TERM_CRITERIA_EPS (criteria tuple at s1:248) */
#define ICELK_CRIT_COUNT 1
#define ICELK_CRIT_EPS 2
/* cv2.OPTFLOW_USE_INITIAL_FLOW / cv2.OPTFLOW_LK_GET_MIN_EIGENVALS */
#define ICELK_FLAG_INITIAL_FLOW 4
#define ICELK_FLAG_MIN_EIGENVALS 8
/* testing aid: force the window-size-generic LK kernel where a specialised one exists (same results) */
#define ICELK_FLAG_GENERIC_KERNEL 0x100
/* the several-features-per-wave form of the specialised kernels (k_lk_multi.hip) instead of the default
 * one-feature-per-wave form: fewer instructions per feature, lower occupancy; same results */
#define ICELK_FLAG_MULTI_PER_WAVE 0x200
/* fixed-point coefficient sets of cv2.cvtColor(COLOR_BGR2GRAY): OpenCV 3.x (14 bit), 4.x (15 bit) */
#define ICELK_GRAY_CV3 3
#define ICELK_GRAY_CV4 4
/* forward-backward distance of icelk_track_fb / icelk_seg_track: np.hypot on float32 as s1:330 (default), or the
 * float32 expression (dx**2 + dy**2)**0.5 of the demo script s0_1:99 */
#define ICELK_FB_HYPOT 0
#define ICELK_FB_SQRT 1

#define ICELK_MAX_LEVELS 12 /* pyramid images per slot (maxLevel <= 11) */

/* ---- library / handle ------------------------------------------------------------------- */
int icelk_version(void);
/* Text of the last error on this handle (or of the last failed icelk_create when h == NULL). */
const char* icelk_last_error(icelk_t* h);
/* One handle per GPU.  max_w/max_h (<= 65535) bound the frame size, n_slots the resident frames,
 * max_pts the features per call (maxCorners / len(tracks)). */
int icelk_create(int device, int max_w, int max_h, int n_slots, int max_pts, icelk_t** out);
int icelk_destroy(icelk_t* h);
/* Run all work on the caller's HIP stream (hipStream_t as void*; NULL = the handle's own stream). */
int icelk_set_stream(icelk_t* h, void* hip_stream);
int icelk_sync(icelk_t* h);
/* Testing / measurement aid: which of the three bit-identical tracker kernels every LK call of this handle uses:
 * 0 = the default choice, ICELK_FLAG_GENERIC_KERNEL, or ICELK_FLAG_MULTI_PER_WAVE. */
int icelk_set_lk_kernel(icelk_t* h, int which);
/* How the fused tracker calls form dist from |p0 - p0r| (ICELK_FB_HYPOT / ICELK_FB_SQRT); the two can differ in
 * the last bit, which flips `valid` for a distance within one ulp of the threshold. */
int icelk_set_fb_distance(icelk_t* h, int form);
/* Named variants of the OpenCV semantics that depend on how OpenCV was BUILT (the oracle names the
 * same switches: oracle/icelk_oracle.c orc_set_variant, and tools/oracle_variants.py measures how far they are apart).
 * 0 is the default of each and what the tuned kernels compute; a non-zero value routes the call through the
 * window-generic tracker kernel / the any-blockSize corner kernel, which carry the variants (slower, same interface):
 *   "lk_sums"    1 | 2   A11, A12, A22, b1, b2 summed in the float lanes of OpenCV 3.x's SSE2 block | 4.x's CV_SIMD128 block
 *                        instead of exactly (at the reference's own parameters 5 of 40 416 features move by more than
 *                        1e-3 px between the forms, none changes status)
 *   "sobel_fma"  bit 0   the Sobel column pass fused (4.x SymmColumnSmallVec_32f in an FMA3 build); bit 1: the row pass fused
 *   "eig_fma"    1       calcMinEigenVal's (a-c)^2 + b^2 as one fused multiply-add
 * (the corner SET is the same under all of them on the test frames, the order of near-equal corners changes.)
 * A cv2 cross-check that finds one of them to be what the reference's OpenCV build does flips this switch. */
int icelk_set_variant(icelk_t* h, const char* name, int value);

/* ---- frame ingest: replaces cv2.cvtColor(frame, cv2.COLOR_BGR2GRAY) at s1:283,311 / s0_1:71,80 */
/* host 8-bit gray image -> slot (level 0); invalidates the slot's pyramid. */
int icelk_upload_gray(icelk_t* h, int slot, const uint8_t* host, int w, int h_, int stride);
/* host 8-bit 3-channel image -> gray in slot.  Channel 0 gets the "B" coefficient: feeding PIL's RGB
 * arrays, as the reference does (s1:310-311), reproduces its swapped weights. */
int icelk_upload_bgr(icelk_t* h, int slot, const uint8_t* host, int w, int h_, int stride, int gray_variant);
/* same two, from device memory (e.g. a torch tensor's data_ptr) */
int icelk_set_gray_device(icelk_t* h, int slot, const void* dev, int w, int h_, int stride);
int icelk_cvt_bgr_device(icelk_t* h, int slot, const void* dev_bgr, int w, int h_, int stride, int gray_variant);
/* asynchronous upload from PINNED host memory on the handle's copy stream (double-buffered
 * streaming, BASELINE.json configs[2]); compute on `slot` waits for the copy by an event. */
int icelk_upload_gray_async(icelk_t* h, int slot, const uint8_t* pinned_host, int w, int h_, int stride);
int icelk_host_alloc(void** out, uint64_t bytes); /* pinned host memory for the call above */
int icelk_host_free(void* p);
/* procedural frame generated on the device (integer value noise, bit-identical to
 * iceberg_tracking_code_amd/synth.py); ux,uy = shift in 1/256 px. */
int icelk_synth_frame(icelk_t* h, int slot, int w, int h_, int64_t ux, int64_t uy, uint32_t seed);
/* the same with a small affine deformation on top of the shift: the texture is sampled at
 * x + ux/256 + (a[0] x + a[1] y) / 2^20,  y + uy/256 + (a[2] x + a[3] y) / 2^20  (|a[k]| <= 2^13, i.e. 0.8 %), so the
 * motion between two frames varies over the frame (shear / scale / rotation); NULL = none. */
int icelk_synth_frame_affine(icelk_t* h, int slot, int w, int h_, int64_t ux, int64_t uy, uint32_t seed,
                             const int32_t* affine);
/* Forget levels >= 1 of a slot whose level 0 stays resident, so the next tracker call rebuilds the
 * pyramid (a frame that is already in HBM re-enters the loop without a copy). */
int icelk_drop_pyramid(icelk_t* h, int slot);
/* read back pyramid level `level` of a slot (level 0 = the gray frame). */
int icelk_download_level(icelk_t* h, int slot, int level, uint8_t* host, int stride, int* w, int* h_);

/* ---- pyramid: cv::buildOpticalFlowPyramid inside cv2.calcOpticalFlowPyrLK (s1:323,326) -------- */
/* Builds levels 1..L of the slot, L = min(max_level, first level whose successor is <= winSize).
 * *out_levels receives L.  icelk_pyrlk / icelk_track_fb call this themselves when needed. */
int icelk_build_pyramid(icelk_t* h, int slot, int win_w, int win_h, int max_level, int* out_levels);
/* The same build enqueued on the handle's copy stream, for a frame that will be tracked LATER: it follows an
 * icelk_upload_gray_async of the slot in stream order and overlaps whatever the compute stream is doing; the
 * next call that needs the pyramid waits for it.  Returns at once. */
int icelk_build_pyramid_ahead(icelk_t* h, int slot, int win_w, int win_h, int max_level);

/* ---- tracker: replaces cv2.calcOpticalFlowPyrLK(img0, img1, p0, None, **lk_params) s1:323,326 - */
/* next_xy is an input as well when flags has ICELK_FLAG_INITIAL_FLOW.  n == 0 is not an error. */
int icelk_pyrlk(icelk_t* h, int prev_slot, int next_slot, const float* prev_xy, float* next_xy,
                uint8_t* status, float* err, int n, int win_w, int win_h, int max_level,
                int crit_type, int max_count, double epsilon, int flags, double min_eig_threshold);
/* Fused forward + backward + distance test of s1:323-333 (one launch, pyramids built once):
 *   p1 = LK(slot0 -> slot1, p0);  p0r = LK(slot1 -> slot0, p1);
 *   dist = np.hypot(|p0 - p0r|) on float32 (s1:329-330; see icelk_set_fb_distance);  valid = dist < fb_threshold.
 * Any output pointer may be NULL. */
int icelk_track_fb(icelk_t* h, int slot0, int slot1, const float* p0, int n, int win_w, int win_h,
                   int max_level, int crit_type, int max_count, double epsilon, double min_eig_threshold,
                   float fb_threshold, float* p1, float* p0r, uint8_t* st_fwd, uint8_t* st_bwd,
                   float* err_fwd, float* err_bwd, float* dist, uint8_t* valid);

/* The filter of s1:329-333 on its own, for callers of the plain icelk_pyrlk: diff = abs(p0 - p0r) in float32,
 * dist = np.hypot(diff) (or the s0_1:99 form, icelk_set_fb_distance), valid = dist < fb_threshold.  Runs the very
 * device function the fused launches end with.  dist / valid may be NULL. */
int icelk_fb_filter(icelk_t* h, const float* p0, const float* p0r, int n, float fb_threshold, float* dist,
                    uint8_t* valid);

/* ---- detector: replaces cv2.goodFeaturesToTrack(frame_gray, mask=mask, **feature_params) s1:437 */
/* Mask (s1:285-294) is uploaded once and reused; NULL clears it. */
int icelk_set_mask(icelk_t* h, const uint8_t* host_mask, int w, int h_, int stride);
/* The mask of s1:285-291 built on the device from the digitised water polygon: poly_xy = n (x, y) pairs on the
 * UNCROPPED photo (`maskpoly`, camtools.py:165); the polygon is shifted by the crop offsets and every pixel centre of
 * the w x h_ frame is tested as Camera.mask_meshgrid does (camtools.py:184-211, matplotlib's contains_points rule,
 * radius 0); inside = 255.  n <= 65536. */
int icelk_set_mask_polygon(icelk_t* h, const double* poly_xy, int n, double crop_left, double crop_top, int w, int h_);
/* Copy the current mask to the host (parity / inspection). */
int icelk_download_mask(icelk_t* h, uint8_t* host_mask, int stride, int* w, int* h_);
/* cornerMinEigenVal map of the slot (debug / parity). */
int icelk_min_eig_map(icelk_t* h, int slot, int block_size, float* host_out, int stride_elems);
/* Shi-Tomasi corners in response order.  max_corners <= 0 = unlimited (up to max_pts).
 * *out_n == 0 corresponds to cv2 returning None (guarded at s1:445). */
int icelk_good_features(icelk_t* h, int slot, int use_mask, int max_corners, double quality_level,
                        double min_distance, int block_size, float* out_xy, int cap, int* out_n);

/* Work counters of the latest detection on this handle: local maxima above the quality threshold, and
 * corners surviving the minDistance rule (before the maxCorners cut). */
int icelk_detect_stats(icelk_t* h, int* n_candidates, int* n_accepted);
/* Work counters of the two-pass candidate stage of the latest detection on a frame_w x frame_h frame (diagnostics; waits
 * for the device): out[0] tiles, [1] pixels listed as possible local maxima, [2] tiles whose list overflowed, [3] entries
 * handed to the 3x3 tie pass, [4] pixels listed as possible carriers of the maximum, [5] tiles whose such list overflowed,
 * [6] longest tile list, [7] listed pixels that got their exact value in the one-pixel pass (certain, above the cut). */
int icelk_detect_fast_stats(icelk_t* h, int frame_w, int frame_h, long long* out);

/* ---- device-resident segment state: the `tracks` / `trackquality` lists of s1:299-300,335-359 --
 * A segment starts at a detection frame (counter % track_len == 0, s1:362,437-448) and is extended
 * by one vertex per tracked frame; only tracks passing the forward-backward test survive, in order.
 * Nothing crosses PCIe until icelk_seg_read. */
int icelk_seg_detect(icelk_t* h, int slot, int use_mask, int max_corners, double quality_level,
                     double min_distance, int block_size, int* out_n);
/* The same detection split in two, for pipelined loops: _begin enqueues the detector on the handle's
 * detection stream (it needs the frame only, so it overlaps a tracker launch issued after it) and returns
 * at once; _finish waits for it and starts the new segment.  TWO detections may be in flight (a third _begin returns
 * ICELK_ESTATE); _finish / _stage take them in the order they were begun and wait for the kernels of that one only.
 * Beginning the detection of frame d+2 before staging the one of frame d lets the host round trip of a detection find
 * kernels that had a whole tracker launch to finish, instead of standing in a serial loop with them. */
int icelk_seg_detect_begin(icelk_t* h, int slot, int use_mask, int max_corners, double quality_level,
                           double min_distance, int block_size);
int icelk_seg_detect_finish(icelk_t* h, int max_corners, int* out_n);
/* Abandon everything that was started ahead and never used: detections begun but not finished (up to two), prepared
 * candidates, a staged segment that was never switched to.  Waits for their kernels, then forgets them; the current
 * segment and every slot stay as they are.  For a loop that announced frames ahead (look-ahead) and ends, or jumps,
 * before they arrive: afterwards the one-call forms (icelk_good_features, icelk_seg_detect) work again. */
int icelk_seg_detect_cancel(icelk_t* h);
/* _finish in two halves, for loops that know their frames some steps ahead: _stage waits for the OLDEST detection in
 * flight and builds the new segment in the handle's next set of segment buffers (four rotate) while the current segment
 * is still being tracked; icelk_seg_switch (no GPU work, no wait) makes the staged segment the current one.  One
 * segment can be staged at a time.  With the detection of frame c begun at step c-4 and staged at step c-2 (behind the
 * _switch of that step), the one host round trip of a detection is off the critical path: the tracker launch of
 * frame c finds its segment ready.  _finish == _stage followed by _switch. */
int icelk_seg_detect_stage(icelk_t* h, int max_corners, int* out_n);
/* _stage that never waits: *out_done = 0 (and nothing done) while the kernels of the oldest detection in flight have
 * not delivered their counts yet, else _stage (*out_done = 1).  A loop that looks ahead calls this at every step from
 * the first one at which the detection may be through, and the waiting form only when the segment is needed at the
 * next step: the host thread -- which also issues the uploads and the tracker launches -- then never stands behind the
 * detector's kernels (with the waiting form it stood there for a third of every period of the PCIe-fed loop). */
int icelk_seg_detect_stage_try(icelk_t* h, int max_corners, int* out_n, int* out_done);
int icelk_seg_switch(icelk_t* h);
/* Optional, ahead of _begin: produce the corner candidates (min-eigenvalue map + non-max test, the part of
 * s1:437 that depends on nothing but the frame and blockSize) of a frame that is already in `slot`, on a stream of
 * its own and into a spare buffer (three exist: two detections in flight + one prepared), while earlier detections are
 * still in their min-distance stage.  A later _begin for the same slot/frame/blockSize/mask adopts the result;
 * otherwise it is dropped.  Results are identical either way. */
int icelk_seg_detect_prepare(icelk_t* h, int slot, int use_mask, int block_size);
int icelk_seg_track(icelk_t* h, int slot_prev, int slot_next, int win_w, int win_h, int max_level,
                    int crit_type, int max_count, double epsilon, double min_eig_threshold,
                    float fb_threshold, int* out_live);
/* tracks: (n, n_vertices, 2) float32; quality: (n, n_vertices-1) float32 -- the arrays np.savez
 * writes at s1:394-395.  cap = rows available in the host buffers, max_vertices = their vertex
 * dimension. */
int icelk_seg_read(icelk_t* h, float* tracks, float* quality, int cap, int max_vertices, int* out_n,
                   int* out_vertices);
/* The same gather into DEVICE memory of the caller (e.g. one slice of a torch tensor that is all-gathered over RCCL at
 * the end of a sharded run, BASELINE.json configs[3]): rows of the surviving tracks, packed, (n, n_vertices, 2) float32
 * at dev_tracks, (n, n_vertices-1) at dev_quality (may be NULL), n as one int32 at dev_count.  cap_rows = rows the
 * buffers hold, must be >= the tracks the segment started with.  Enqueued on the handle's compute stream: no wait, no
 * host read-back. */
int icelk_seg_archive(icelk_t* h, void* dev_tracks, void* dev_quality, void* dev_count, int cap_rows, int* out_vertices);
/* The LAST pair of a segment and the FIRST pair of the next one are independent (s1:362 tracks the old features across
 * (c-1, c), s1:440 starts the new segment from the corners of frame c, tracked across (c, c+1) one loop pass later).
 * icelk_seg_track_defer takes the arguments of icelk_seg_track_async but launches nothing: the pair waits, and the next
 * icelk_seg_track_async / _defer after icelk_seg_switch sends both pairs to the device as ONE tracker launch (ramp-up
 * and tail of the launch are paid once; every workgroup still tracks one feature exactly as before).  The waiting pair
 * goes out on its own whenever its result or its frames are needed first: icelk_seg_flush, icelk_sync, a read-out of
 * its segment, a second switch, or an ingest / icelk_drop_pyramid into one of its two slots; also when the partner's
 * LK parameters differ.  Results are those of icelk_seg_track_async in every case.
 * After icelk_seg_switch the segment it closed stays addressable until the switch after: the _closed forms of the
 * read-outs gather from it (and launch its waiting pair first if it still waits). */
int icelk_seg_track_defer(icelk_t* h, int slot_prev, int slot_next, int win_w, int win_h, int max_level,
                          int crit_type, int max_count, double epsilon, double min_eig_threshold,
                          float fb_threshold);
int icelk_seg_flush(icelk_t* h);
/* Pairs per segment of the loop that drives this handle (`track_len` of s1_lucaskanade_tracking.py:128,362); 0 = unknown
 * (the default).  A hint, results never depend on it: the backward pass of pair v of a segment builds, level by level,
 * exactly the templates (patch, derivatives, 2x2 matrix) that the forward pass of pair v+1 builds again -- frame v+1 at
 * the positions the tracks have reached -- so the window-specialised kernels leave them in HBM for the next launch
 * instead.  With the hint the LAST pair of a segment does not write templates nobody will read.
 * ICELK_NO_TEMPLATE_REUSE=1 turns the reuse off altogether (A/B). */
int icelk_seg_track_len_hint(icelk_t* h, int track_len);
/* out[0] = segment pairs of this handle whose forward pass took its templates from the pair before, out[1] = pairs whose
 * backward pass left templates for a successor (diagnostics; tests/test_gpu_api.py uses it to see the reuse engage and
 * refuse: a pair whose first frame is not the frame the templates were built on builds its own). */
int icelk_seg_template_stats(icelk_t* h, long long* out);
/* The template tables behind icelk_seg_track_len_hint: bytes of ONE of the two tables, the tracks (rows) it has room for,
 * and state 0 = in use (or not needed yet), 1 = switched off (ICELK_NO_TEMPLATE_REUSE), 2 = off because the allocation
 * failed.  They are allocated once per row geometry (window, pyramid levels) at the first pair that can use them, for
 * max_pts rows within ICELK_TEMPLATE_BUDGET_MB (default 8192 for both tables; at most half of the free device memory); a
 * segment with more tracks than rows builds its own templates. */
int icelk_seg_template_info(icelk_t* h, long long* bytes_per_table, long long* rows, int* state);
/* out[0] = segments whose tables were written by the device-driven tail of their detection (k_tail.hip: sort, maxCorners
 * cut, corner list = the reset `tracks = [[(x, y)] ...]` of s1:440-448, launch order -- all from the device-side counts,
 * enqueued by icelk_seg_detect_begin; icelk_seg_detect_stage only adopts the verdict), out[1] = segments staged by the
 * host's tail (min-distance relaxation not converged, a pruned candidate set that fell short of maxCorners,
 * minDistance < 1, or ICELK_HOST_TAIL=1).  Results are the same either way. */
int icelk_seg_tail_stats(icelk_t* h, long long* out);
int icelk_seg_read_closed(icelk_t* h, float* tracks, float* quality, int cap, int max_vertices, int* out_n,
                          int* out_vertices);
int icelk_seg_archive_closed(icelk_t* h, void* dev_tracks, void* dev_quality, void* dev_count, int cap_rows,
                             int* out_vertices);
/* non-blocking variants for pipelined loops: no host read-back, counts stay on the device */
int icelk_seg_track_async(icelk_t* h, int slot_prev, int slot_next, int win_w, int win_h, int max_level,
                          int crit_type, int max_count, double epsilon, double min_eig_threshold,
                          float fb_threshold);
int icelk_seg_live(icelk_t* h, int* out_live, int64_t* out_tracked_total);

/* ---- projection of finished tracks to map coordinates + plausibility filter (the consumer of the path) --
 * Replaces the per-track Python loops of s2_cam_to_utm.py:243-347: every vertex is moved to uncropped photo
 * coordinates (imports/camtools.py:414-421) and projected onto the sea-level plane (Camera.photo_to_utm,
 * imports/camtools.py:286-332); u, v [m/s] = vertex difference / interval, speed = hypot(u, v); a track is
 * dropped when mean(speed) < min_speed or max(speed) > max_speed, or -- if max(speed) > speed_threshold -- when
 * consecutive vectors differ by more than max_speedfactor in speed or max_angle degrees in direction.
 * float64 throughout, the reference's operation order.  X, U, V are the direction cosines of
 * camtools.py:300-316, formed by the caller (utm.py does it with numpy, as the reference does). */
typedef struct icelk_camera {
    double X[3], U[3], V[3];
    double sigma;              /* image_width / sensor_width * sigma   (camtools.py:145) */
    double H, E, N;            /* camera height above the water (tide corrected), easting, northing */
    double half_w, half_h;     /* pic['width'] / 2.0, pic['height'] / 2.0 of the UNCROPPED photo */
    double crop_left, crop_top;
} icelk_camera_t;
typedef struct icelk_utm_filter {
    double interval_s;         /* seconds between consecutive vertices (the `_at_{dt}sec_` of the file name) */
    double max_speed, min_speed, max_speedfactor, max_angle, speed_threshold;   /* s2_cam_to_utm.py:84-88 */
} icelk_utm_filter_t;
/* tracks: host (n, n_vertices, 2) float32, the `tracks` array of one .npz (s1:394-395).  x, y, u, v, speed: host
 * (n, n_vertices-1) float64, x/y = map position of each vector's FIRST vertex (s2:296-297).  keep: n bytes --
 * 1 kept, 0 dropped, 2 = the reference raises ValueError on this track (max() of an empty list: fewer than two
 * vectors and faster than speed_threshold).  n <= max_pts, n_vertices <= 17. */
int icelk_project_tracks(icelk_t* h, const float* tracks, int n, int n_vertices, const icelk_camera_t* cam,
                         const icelk_utm_filter_t* filt, double* x, double* y, double* u, double* v, double* speed,
                         uint8_t* keep);
/* The same for the segment held on the device (icelk_seg_*): gather of the surviving tracks + projection, only
 * results cross PCIe.  cap = rows of the host buffers, max_vectors = their second dimension. */
int icelk_seg_project(icelk_t* h, const icelk_camera_t* cam, const icelk_utm_filter_t* filt, int cap, int max_vectors,
                      double* x, double* y, double* u, double* v, double* speed, uint8_t* keep, int* out_n,
                      int* out_vectors);

/* ---- gridding of the projected velocities (s3_utm_to_gridded_utm.py:391-421) --------------------------------
 * matplotlib.path.Path(poly).contains_points(points) with radius 0 (the rule of icelk_set_mask_polygon) for arbitrary
 * float64 points: used for "does the fjord outline contain the cell centre" (imports/tracking_misc.py:49). */
int icelk_points_in_polygon(icelk_t* h, const double* poly_xy, int n_poly, const double* pts_xy, int n_pts,
                            uint8_t* inside);
/* Square cells of `spacing` from (left, top), cols x rows, cell (i, j) = column i, row j downwards, stored at
 * i * rows + j as the reference walks them (tracking_misc.py:41-43); cell_on[] marks the cells the grid keeps.  For
 * every kept cell: count of the n velocities (x, y, u, v; float64) whose position lies in it by contains_points'
 * rule, and for count > 0 mean_u = np.sum(u_sel) / count (numpy's pairwise order), mean_v, speed = hypot. */
int icelk_grid_bin(icelk_t* h, const double* x, const double* y, const double* u, const double* v, int n, double left,
                   double top, double spacing, int cols, int rows, const uint8_t* cell_on, int* count, double* mean_u,
                   double* mean_v, double* speed);

/* ---- measurement ------------------------------------------------------------------------------ */
/* Per-kernel HIP-event timing on the handle's streams (bench.py's roofline leg).  on = 1: every kernel; on = 2: the
 * tracker launches only (each timed kernel costs two event records on its stream, which the chains of short detector
 * kernels feel); on = 0: off, durations collected so far are added up. */
int icelk_prof_enable(icelk_t* h, int on);
int icelk_prof_reset(icelk_t* h);
/* LK iterations every feature of the latest tracker call ran while profiling was enabled: forward pass in the low 16
 * bits, backward pass in the high 16 (0xffffffff = a track that was already dead); *out_n = features of that call.
 * The iterations-per-feature histogram of bench.py comes from here. */
int icelk_prof_iterations(icelk_t* h, uint32_t* host_out, int cap, int* out_n);
/* Outcome of the hardware-queue probe icelk_create ran for this handle: picks[0..3] = which of the eight
 * candidate streams became the detection / candidates / pyramid / tail stream (-1: ICELK_NO_STREAM_PROBE, creation
 * order); *quickest = the smallest fraction of the filler's duration after which a one-wave kernel on a candidate stream
 * came back beside a busy compute stream, *limit = the fraction above which a stream counted as held up.  Throughput
 * depends on these picks, results do not: bench.py records them with every line. */
int icelk_stream_probe_info(icelk_t* h, int* picks, double* quickest, double* limit);
int icelk_prof_count(void);
const char* icelk_prof_name(int kernel_id);
int icelk_prof_get(icelk_t* h, int kernel_id, int* launches, double* total_ms);

#ifdef __cplusplus
}
#endif
#endif /* ICELK_H */
