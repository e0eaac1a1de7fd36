// icelk_internal.h -- shared between the translation units of libicelk.so (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/icelk.h"

namespace icelk {

constexpr int kMaxLevels = ICELK_MAX_LEVELS;
constexpr int kPitchAlign = 64;  // bytes; every level row starts 64-B aligned (dword / dwordx4 loads)

// One pyramid level in device memory (unpadded; borders are reflected inside the kernels).
struct Level {
    uint8_t* ptr;
    int w, h, pitch;
};

// By-value kernel argument: the whole pyramid of one frame.
struct Pyramid {
    Level lv[kMaxLevels];
};

struct LKParams {
    int win_w, win_h;
    int top_level;      // effective maxLevel (levels used = top_level + 1)
    int max_count;      // clamped criteria.maxCount
    double eps2;        // clamped criteria.epsilon squared
    int flags;
    float min_eig_thr;
    float fb_thr;
    int margin;         // search-region margin R of the LDS-staged J tile
    float eps2_lo, eps2_hi;   // float values below / above which (double)dx*dx + (double)dy*dy <= eps2 is decided
    int dist_form;      // forward-backward distance: 0 = np.hypot on float32 (s1:330), 1 = (dx^2+dy^2)^0.5 (s0_1:99)
    // how A11, A12, A22, b1, b2 are summed (icelk_set_variant "lk_sums"): 0 = exactly (int64), one rounding; 1 / 2 = in the
    // float lanes of OpenCV 3.x's SSE2 block / 4.x's CV_SIMD128 block (oracle/icelk_oracle.c orc_set_variant).  Non-zero
    // runs in the window-generic kernel only.
    int sum_mode;
};

// kernel ids for the profiling table
enum KernelId {
    K_GRAY = 0,
    K_PYRDOWN,
    K_LK,
    K_LK_FB,
    K_EIG,
    K_SUPPRESS,
    K_EMIT,
    K_PROJECT,
    K_SYNTH,
    K_LK_FB_PAIR,   // two segment pairs in one launch (icelk_seg_track_defer)
    K_COUNT_
};

struct Slot {
    uint8_t* base = nullptr;  // one allocation holding all levels
    size_t bytes = 0;
    int w = 0, h = 0;
    int levels_built = 0;     // number of valid images (0 = empty, 1 = level 0 only)
    Level lv[kMaxLevels];
    hipEvent_t ready = nullptr;   // upload-complete event (async uploads)
    bool pending = false;
    hipEvent_t frame_ev = nullptr;  // level 0 written (recorded on the compute stream by every ingest call)
    // last readers of the slot: pyramid/tracker launches on the compute stream, the corner kernel on the
    // detection stream.  An asynchronous upload into the slot waits for exactly these, not for everything that
    // happens to be queued, so frame t+1 crosses PCIe while frame t is being tracked.
    // `used` refers to the event that covers the latest such launch: the slot's own (`used_own`) or one shared by
    // everything a tracker launch touched (Ctx::launch_ev), so that a launch costs one event record, not one per slot
    hipEvent_t used = nullptr, used_own = nullptr, det_used = nullptr;
    hipEvent_t eig_used = nullptr;   // the corner kernel of icelk_seg_detect_prepare (candidates stream) reads level 0
    unsigned long long gen = 0;   // bumped whenever a new frame enters the slot
};

struct ProfEvt {
    hipEvent_t a, b;
    int id;
};

struct Ctx;  // full definition in icelk_abi.hip

// ---- launchers (each defined in its kernel TU) -------------------------------------------------
void launch_bgr2gray(hipStream_t s, const uint8_t* src, int src_pitch, uint8_t* dst, int dst_pitch,
                     int w, int h, int variant);
void launch_pyrdown(hipStream_t s, const Level& src, const Level& dst);
// lv[first+1 .. first+n] (n = 1..3) from lv[first] in ONE launch (k_pyramid.hip)
void launch_pyramid_fused(hipStream_t s, const Level* lv, int first, int n, bool one_wave = false);
void launch_synth(hipStream_t s, const Level& dst, int64_t ux, int64_t uy, uint32_t seed, const int* affine);

// LK.  p_in/p_out etc. are device pointers.  fb = fused forward+backward.
struct LKBuffers {
    const float* p_in;   // (n,2)
    float* p_fwd;        // (n,2) forward result (in/out when INITIAL_FLOW)
    uint8_t* st_fwd;
    float* err_fwd;
    float* p_bwd;        // fb only
    uint8_t* st_bwd;
    float* err_bwd;
    float* dist;
    uint8_t* valid;
    const int* n_dev;    // optional device-side count (overrides n when non-null)
    const int* order;    // optional launch order (k_tracks.hip k_seg_order); results do not depend on it
    int order_plain;     // walk the order table linearly instead of dealing it to the XCDs (dense features)
    const int* order_border;   // number of leading table entries that are border features: launched first, undealt
    // Segment mode (icelk_seg_track): the launch itself keeps the track table.  Feature f is track f of the
    // segment; dead tracks (seg_alive[f] == 0) exit at once, survivors of the forward-backward test get their
    // new vertex and distance appended and their position updated in place -- the Python loop of
    // s1_lucaskanade_tracking.py:335-359 without a separate compaction launch.
    uint8_t* seg_alive;
    float* seg_xy;               // (n,2) current position of every track; also the input points in this mode
    float* seg_tracks;           // [track][max_vert][2]
    float* seg_quality;          // [track][max_vert-1]
    int seg_vert, seg_max_vert;  // vertex written by this launch
    unsigned long long* seg_tracked;   // 64 sharded counters: features tracked (for throughput accounting)
    // diagnostics (ICELK_LK_STAMPS=<file>): per workgroup {s_memtime at entry, at exit, HW_ID | XCC_ID << 32}
    unsigned long long* stamps;
    // measurement (icelk_prof_enable): LK iterations each feature ran, forward pass in the low half, backward in the high
    uint32_t* iters;
    // Template reuse across the pairs of a segment (window-specialised kernels only; see k_lk_fast.hip): the BACKWARD pass
    // of pair v builds, at every level, the template of frame v+1 at the position the forward pass of pair v+1 starts
    // from -- the same patch, derivatives and 2x2 matrix.  tmpl_out: where the backward pass leaves them; tmpl_in: where
    // the forward pass of this launch finds the ones of the launch before (null: built here).
    // Layout: [track][level][quad][lane] 16-byte pieces; the dword behind a lane's template carries A11 / A12 / A22 (lanes 0-2).
    void* tmpl_out;
    const void* tmpl_in;
    int tmpl_levels;     // levels per track in those tables (top_level + 1 of the launch that wrote them)
};
// 16-byte pieces per lane and level of a stored template for this window (0: no window-specialised kernel, no reuse)
int lk_template_quads(int win_w, int win_h);
// will launch_lk / launch_lk_pair run the window-specialised one-feature-per-wave kernel for these parameters?
bool lk_fast_eligible(const LKParams& P);
// search-tile margin of the window-specialised tracker kernels (lk_fast_tiles.h); the host needs it to tell which features'
// tiles reach over the frame border
constexpr int kLkTileMargin = 1;
// one tracker job: a frame pair and the features tracked across it
struct LKJob {
    Pyramid I, J;
    LKBuffers B;
    int n;
};
size_t lk_lds_bytes(const LKParams& P);
int launch_lk(hipStream_t s, const Pyramid& I, const Pyramid& J, const LKBuffers& B, int n, const LKParams& P,
              bool fb);
// Two INDEPENDENT jobs with the same LK parameters in one launch (the last pair of a closing segment and the first pair
// of the next one: s1:362,440 make them independent): ramp-up and tail of the launch are paid once for both.  Returns
// false when no kernel for this window takes two jobs (the caller then launches them one after the other).
bool launch_lk_pair(hipStream_t s, const LKJob& a, const LKJob& b, const LKParams& P);

// Detector stages.
struct DetectScratch {
    float* eig;            // w*h
    unsigned* max_key;     // 1 (order-preserving key of the masked maximum)
    unsigned long long* cand;   // candidate keys (value bits << 32 | raster index)
    int* cand_count;       // 1
    int cand_cap;
    int* cell_count;       // grid cells + 1
    int* cell_start;       // grid cells + 1
    int* cell_fill;        // grid cells
    int* chunk_tot;        // candidates per 2048 cells (offsets of the scan's workgroups)
    unsigned long long* cell_cand;  // candidates grouped by cell
    uint8_t* state;        // per grouped candidate: 0 undecided, 1 accepted, 2 rejected
    int* undecided;        // 1
    unsigned long long* acc;   // accepted keys
    unsigned long long* acc_sorted;
    unsigned long long* raw;   // candidate regions written by the corner kernel (one of two buffers, see icelk_abi.hip)
    int* acc_count;        // 1
    void* sort_tmp;
    size_t sort_tmp_bytes;
    int* blk_count;        // candidates per producer workgroup (region layout, see k_corners.hip)
    unsigned* key_hist;    // histogram of the response keys (top-K pruning); 65536 entries allocated
    unsigned* prune_key;   // 1: candidates with a smaller response key are ignored (0 = none)
    int src_nblk, src_region;   // geometry of the candidate regions of the detection in flight
    // two-pass detector (k_corners_fast.hip): what its integer pass hands to its exact passes -- per tile the pixels that can
    // be a local maximum {y << 16 | x, upper bound} and those that can carry the maximum, the tile's largest upper bound,
    // and the largest lower bound of the whole frame.  These belong to the candidate buffer (Ctx::EigOut), like raw.
    uint2* acand;
    int* acount;
    uint2* amaxc;
    int* amaxn;
    float* aemax;
    unsigned* fmax_key;      // [0] largest lower bound of the frame, [1] entries in aties
    unsigned* aties;         // pixels (y << 16 | x) whose neighbourhood the exact pass must look at; 0x80000000 | tile: whole tile
    // device-driven tail (k_tail.hip): control words (TailCtl), response-bin histogram / offsets / cursors, histogram of the launch order
    int* tail_ctl;
    int* tail_resp;
    int* tail_bins;
};
// k_tail.hip: words of DetectScratch::tail_ctl
enum TailCtl { TC_TICKET_GATHER = 0, TC_TICKET_RANK, TC_TICKET_ORDER, TC_NOUT, TC_STATUS, TC_CAND, TC_ACC, TC_UNDECIDED, TC_PRUNED, TC_WORDS_ = 16 };
// verdict of the device-driven tail (h_counts[5]): anything but TAIL_OK leaves the tail to the host (detect_finish)
enum TailStatus { TAIL_OK = 0, TAIL_UNDECIDED = 1, TAIL_PRUNED_SHORT = 2, TAIL_OVERFLOW = 3, TAIL_SKEWED = 4 };
constexpr int kTailOrderBins = 8192;
constexpr int kTailRespBins = 4096;
struct TailOrderGeo {   // cells of the launch order (k_tracks.hip k_seg_order): bin 0 = border features
    int cw_shift, ch_shift, cells_x, ncells, w, h, border_px;
};
struct TailReset {      // the counters a detection leaves zeroed for the next one of its set (k_corners.hip k_detect_reset)
    int* cell_count;
    int* cell_fill;
    int ncell;
    int* chunk_tot;
    int* undecided;
    int* acc_count;
    int* cand_count;
    unsigned* key_hist;
    unsigned* prune_key;
    int scan_chunk, chunk_stride, key_bins;
};
TailOrderGeo tail_order_geometry(int w, int h, int border_px);
size_t tail_resp_words();
TailReset tail_reset_of(DetectScratch& D, int ncell);
// the accepted candidates -> D.acc (what k_gather_accepted does) + their response-bin histogram, bin offsets and the verdict
void launch_tail_gather(hipStream_t s, DetectScratch& D, int ncell, double quality, int undecided_index, int max_corners,
                        int cap, int force_status);
// rank + corner tables of the new segment + launch order + counter reset + counts to the host, all from device-side counts
void launch_tail_device(hipStream_t s, DetectScratch& D, double quality, float* seg_xy, uint8_t* seg_alive, float* seg_tracks,
                        int max_vert, int* order, int* order_border, const TailOrderGeo& geo, const TailReset& rs,
                        int* host_counts, int counts_seq_word, int seq);
void launch_min_eig(hipStream_t s, const Level& img, int block_size, float* eig, const uint8_t* mask,
                    int mask_pitch, unsigned* max_key, int variant = 0);
bool fused_block_size(int bs);
// mode: 0 = the cell grid and the counters; | 1 = the key histogram too; | 2 = and the masked maximum D.max_key (only where
// the candidate buffer behind it is known to be this detection's: it may have been handed on since)
void launch_detect_reset(hipStream_t s, DetectScratch& D, int ncell, int mode);
// K6+K7: local maxima into per-workgroup regions of D.raw (stream order, no host sync)
// beside_tracker: the launch is expected to share the device with a tracker launch (one-wave workgroups, k_corners.hip)
void launch_candidates(hipStream_t s, DetectScratch& D, const Level& img, int block_size, const uint8_t* mask,
                       int mask_pitch, double quality, bool use_generic, float* eig_out_or_null, int variant = 0,
                       bool beside_tracker = false);
// K6 + K7 in two passes for blockSize 3 / 5 / 7 / 10 (integer bracket of the map, exact arithmetic at the possible maxima
// only): same regions, counts and max_key as launch_candidates' one-pass kernel.  quality <= 0: no threshold cut.
bool launch_candidates_fast(hipStream_t s, DetectScratch& D, const Level& img, int block_size, const uint8_t* mask,
                            int mask_pitch, double quality);
size_t fast_tiles(int w, int h);
size_t fast_cand_entries(int w, int h);
size_t fast_max_entries(int w, int h);
size_t fast_key_capacity(int w, int h);
size_t fast_regions(int w, int h);
size_t candidate_capacity(int w, int h);   // keys the region layout needs
size_t candidate_blocks(int w, int h);
// minDistance < 1: candidates above the threshold, flat in D.cand / D.cand_count
void launch_flatten(hipStream_t s, DetectScratch& D, double quality);
// K8: regions -> D.acc / D.acc_count (unsorted accepted keys); candidates counted in D.cell_start[ncell];
// D.undecided[suppress_launch_count()-1] != 0 afterwards means the relaxation has not converged yet: call
// continue_min_distance and look again.
// prune_want > 0: only the ~prune_want strongest candidates take part (valid iff the accepted ones reach maxCorners)
// no_gather: the caller gathers the accepted candidates itself (launch_tail_gather: the device-driven tail)
void launch_min_distance(hipStream_t s, DetectScratch& D, int w, int h, double min_distance, double quality,
                         int prune_want, bool no_gather = false);
void continue_min_distance(hipStream_t s, DetectScratch& D, int w, int h, double min_distance);
int suppress_launch_count();
size_t sort_tmp_bytes(int n);
void sort_keys_desc(hipStream_t s, DetectScratch& D, const unsigned long long* in, unsigned long long* out,
                    int n);
void launch_emit_corners(hipStream_t s, const unsigned long long* keys, int n, int w, float* xy);
// corner list + the new segment's tables + the detector set's counter reset in one launch (k_tail)
void launch_tail_fused(hipStream_t s, const unsigned long long* keys, int n, float* corners, float* seg_xy,
                       uint8_t* seg_alive, float* seg_tracks, int max_vert, DetectScratch& D, int ncell, int mode);

// Segment bookkeeping (k_tracks.hip).
void launch_fb_filter(hipStream_t s, const float* p0, const float* p0r, int n, float thr, int form, float* dist,
                      uint8_t* valid);
void launch_seg_init(hipStream_t s, const float* corners, int n, float* xy, uint8_t* alive, float* tracks,
                     int max_vert);
// {alive tracks, features tracked so far} -> host_out[0..1] (pinned, 64-bit each)
// projection epilogue (k_utm.hip); the structs are the public ones
typedef icelk_camera_t UtmCamera;
typedef icelk_utm_filter_t UtmFilter;
void launch_project_tracks(hipStream_t s, const float* tracks, int n, int nv, const UtmCamera& cam, const UtmFilter& f,
                           double* x, double* y, double* u, double* v, double* speed, uint8_t* keep);
void launch_points_in_polygon(hipStream_t s, const double* poly, int n, const double* pts, int m, uint8_t* out);
void launch_grid_assign(hipStream_t s, const double* x, const double* y, int n, double left, double top, double spacing,
                        int cols, int rows, const uint8_t* cell_on, unsigned long long* keys, int* key_count, int key_cap);
void launch_grid_reduce(hipStream_t s, const unsigned long long* keys, const int* key_count, const double* u,
                        const double* v, int ncells, int* count, double* mean_u, double* mean_v, double* speed);
size_t sort_keys_asc(hipStream_t s, void* tmp, size_t tmp_bytes, const unsigned long long* in, unsigned long long* out,
                     int n, int end_bit);
void launch_polygon_mask(hipStream_t s, const double* poly, int n, double crop_left, double crop_top, int w, int h,
                         uint8_t* mask, int pitch);
void launch_seg_order(hipStream_t s, const float* xy, int n, int w, int h, int border_px, int* order, int* border_count);
void launch_seg_stats(hipStream_t s, const uint8_t* alive, int n, const unsigned long long* tracked_shards,
                      unsigned long long* host_out);
// rows of the alive tracks, in track order, packed into out_tracks (n_alive, nvert, 2) / out_quality
void launch_seg_gather(hipStream_t s, const uint8_t* alive, int n, const float* tracks, const float* quality,
                       int nvert, int max_vert, float* out_tracks, float* out_quality, int* out_count = nullptr);
size_t min_eig_lds_bytes(int block_size);

}  // namespace icelk
