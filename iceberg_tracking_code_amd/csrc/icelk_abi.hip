// icelk_abi.hip -- handle, device memory and the extern "C" entry points declared in include/icelk.h.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <dlfcn.h>

#include <algorithm>
#include <mutex>

#include "icelk_internal.h"

namespace icelk {

constexpr int kSegSets = 6;
constexpr int kLaunchEvents = 32;
constexpr int kMaxVert = 17;  // vertices per track kept on the device (track_len <= 16; reference uses 2)

static const char* kKernelNames[K_COUNT_] = {
    "bgr2gray", "pyrdown", "lk", "lk_fb", "corner_candidates", "min_distance", "sort_emit",
    "project_tracks", "synth", "lk_fb_pair",
};

struct DetectJob {
    bool active = false;
    int w = 0, h = 0;
    double quality = 0, min_distance = 0;
    size_t ncell = 0;
    const int* cand_count_ptr = nullptr;
    int prune_want = 0;   // > 0: top-K pruning is on for this job
    unsigned long long seq = 0;   // order of icelk_seg_detect_begin calls: the oldest job in flight is finished first
    // device-driven tail (k_tail.hip): enqueued behind the min-distance stage by detect_begin; the corners of this detection
    // start a segment in set `seg_set`, cut at max_corners
    bool dev_tail = false;
    int seg_set = -1;
    int max_corners = 0;
};

struct Ctx {
    int device = 0;
    int max_w = 0, max_h = 0, n_slots = 0, max_pts = 0;
    hipStream_t own_stream = nullptr, stream = nullptr, copy_stream = nullptr;
    // Asynchronous uploads alternate between two streams: between two copies of ONE stream the runtime spends ~50 us
    // (completion signal of the first, dependency of the second: 220 us copies came out 270 us apart), which a copy
    // queued on the other stream fills
    hipStream_t copy_stream2 = nullptr;
    // the streams of icelk_upload_gray_async since round 4: created at the first such upload, HIGH priority -- a stream of
    // the compute stream's class can share its hardware queue, and then the copy's dependencies wait behind a tracker launch
    hipStream_t copy_hi[2] = {nullptr, nullptr};
    hipStream_t copy_more[2] = {nullptr, nullptr};   // ICELK_COPY_STREAMS=3|4 (A/B measurements)
    int n_copy_streams = 2;
    unsigned upload_seq = 0;
    // pyramids built ahead of their step: not on the copy stream, where a 12 MB upload of a LATER frame would stand
    // between a pyramid and the tracker launch that waits for it
    hipStream_t pyr_stream = nullptr;
    int side_pick[4] = {-1, -1, -1, -1};   // which of the probed candidate streams became detection / candidates / pyramid / tail
    double probe_limit = 0, probe_quickest = 0;
    // Detection (corner candidates, min-distance, sort) runs on its own stream: it only needs the frame,
    // not the tracker's results, so it overlaps the LK launch of the same frame (s1:323-326 vs s1:437).
    hipStream_t det_stream = nullptr;
    hipEvent_t det_done = nullptr;      // corners of the latest detection are in d_corners
    hipEvent_t corners_free = nullptr;  // the compute stream has consumed d_corners
    std::vector<Slot> slots;
    std::string err;

    // staging for 3-channel uploads
    uint8_t* d_bgr = nullptr;
    int bgr_pitch = 0;
    // detector mask
    uint8_t* d_mask = nullptr;
    int mask_pitch = 0;
    bool has_mask = false;
    int mask_w = 0, mask_h = 0;

    // point buffers
    float *d_p0 = nullptr, *d_p1 = nullptr, *d_p0r = nullptr, *d_err_f = nullptr, *d_err_b = nullptr,
          *d_dist = nullptr, *d_corners = nullptr;
    uint8_t *d_st_f = nullptr, *d_st_b = nullptr, *d_valid = nullptr;

    DetectScratch D{};
    size_t ncell_cap = 0;
    // Output of the corner kernel (candidate regions, per-tile counts, masked maximum), double buffered: the
    // candidates of a FUTURE detection frame can be produced (icelk_seg_detect_prepare, on eig_stream) while the
    // min-distance stage of the current one still reads its own.  D.raw / D.blk_count / D.max_key /
    // D.src_* always mirror eo[eo_active].
    struct EigOut {
        unsigned long long* raw = nullptr;
        int* blk_count = nullptr;
        unsigned* max_key = nullptr;
        // scratch of the two-pass detector (DetectScratch::acand ...)
        uint2* acand = nullptr;
        int* acount = nullptr;
        uint2* amaxc = nullptr;
        int* amaxn = nullptr;
        float* aemax = nullptr;
        unsigned* fmax_key = nullptr;
        unsigned* aties = nullptr;
        double quality = 0;          // candidates below max * quality were never given their exact key (0: none were cut)
        int nblk = 0, region = 0;
        bool valid = false;          // holds the candidates of (slot, gen) for (block_size, use_mask, mask_gen)
        int slot = -1, block_size = 0, use_mask = 0;
        unsigned long long gen = 0, mask_gen = 0;
        hipEvent_t done = nullptr;
    } eo[3];
    int eo_active = 0;
    // Two detections may be in flight (begun, not finished): the min-distance stage of frame d+2 is issued before the
    // host round trip of frame d, so that the round trip finds kernels that had a whole tracker launch to finish
    // instead of standing in a serial loop with them.  The scratch of a detection (D, job, h_counts, the flags below,
    // eo_active) exists twice; the members of this struct proper are the WORKING COPY of set `dset_cur`
    // (det_load / det_save), which keeps every detector function written against one set.
    struct DetSet {
        DetectScratch D{};
        DetectJob job{};
        int* h_counts = nullptr;
        int counts_seq = 0;               // h_counts[kCountsSeq] == counts_seq: the counts published last have arrived
        hipEvent_t counts_ev = nullptr;   // h_counts holds the counts of this set's detection
        hipEvent_t tail_done = nullptr;   // the tail of this set's latest detection (sort, emit, counter reset) is through
        size_t reset_ncell = 0;
        bool counters_clean = false;
        int eo_active = 0;
    } dset[2];
    hipEvent_t counts_ev = nullptr, tail_done = nullptr;
    int counts_seq = 0;
    hipStream_t tail_stream = nullptr;     // see detect_finish
    int dset_cur = 0;
    unsigned long long job_seq = 0;
    hipStream_t eig_stream = nullptr;
    unsigned long long mask_gen = 0;
    double prep_quality = 0;   // qualityLevel of the latest detection begun: what icelk_seg_detect_prepare cuts its candidates at

    // segment state
    // Segment state, two sets: while the tracker launch of a detection frame still extends the closing segment in
    // one set, the new segment is initialised in the other (on the detection stream, right after the corners are
    // emitted) -- the next tracker launch does not have to wait for an initialisation queued behind its predecessor.
    struct SegBuf {
        float* live = nullptr;      // (max_pts,2) current position of every track of the segment
        uint8_t* alive = nullptr;   // 1 while the track survives
        int* order = nullptr;       // spatial launch order of the segment's tracks (k_seg_order)
        int* order_border = nullptr;   // 1 int: leading entries of `order` that are border features
        float* tracks = nullptr;    // [track][kMaxVert][2]
        float* quality = nullptr;   // [track][kMaxVert-1]
        // last launch on the compute stream that touches this set: own event or a shared launch event (see Slot::used)
        hipEvent_t used = nullptr, used_own = nullptr;
        hipEvent_t ready = nullptr;   // the tables of the segment staged in this set are written (detection or tail stream)
        int vert = 0, upper = 0;    // vertices so far, tracks of the segment (= corners detected)
        // templates the backward pass of the latest pair left in tmpl.buf[set & 1] serve the forward pass of the pair that
        // writes vertex `tmpl_for` (with the window / levels of tmpl_key); -1: none
        int tmpl_for = -1, tmpl_key = 0, tmpl_slot = -1;
        unsigned long long tmpl_gen = 0;
    } sb[kSegSets];
    // Template reuse between the pairs of a segment (LKBuffers::tmpl_out; k_lk_fast.hip).  Consecutive segments use
    // consecutive sets, and at most two segments have launches in flight: two tables, picked by the parity of the set.
    struct {
        void* buf[2] = {nullptr, nullptr};
        size_t bytes = 0;          // per table
        size_t row_bytes = 0;      // bytes per track the tables were laid out for (levels x quads x 64 lanes x 16 B)
        size_t budget = (size_t)8 << 30;   // both tables together (ICELK_TEMPLATE_BUDGET_MB)
        int quads = 0, levels = 0;
        bool off = false;          // ICELK_NO_TEMPLATE_REUSE, or the tables could not be allocated
        bool failed = false;       // ... the latter
        long long taken = 0, left = 0;   // pairs whose forward pass took templates / whose backward pass left them
    } tmpl;
    int track_len_hint = 0;        // icelk_seg_track_len_hint: pairs per segment (0: unknown -- every pair leaves templates)
    // Six sets rotate: the current segment, the one staged for the next switch (sb_cur + 1), the one closed by the
    // latest switch (sb_cur - 1), whose last pair may still be waiting (icelk_seg_track_defer) and whose tracks stay
    // readable (icelk_seg_archive_closed) until the switch after, the one before that, which a tracker launch
    // may still be working on when the host, a launch ahead of the device, stages the next segment -- and, since the tail
    // of a detection writes the new segment's tables without the host (k_tail.hip), the sets behind the staged one that
    // the detections in flight (two at most) have reserved (DetectJob::seg_set).
    hipEvent_t launch_ev[kLaunchEvents] = {nullptr};   // one per tracker launch, round robin
    int launch_seq = 0;
    int sb_cur = 0;
    bool closed_valid = false;
    struct Deferred {
        bool pending = false;
        int set = 0, slot_prev = 0, slot_next = 0;
        LKJob job{};
        LKParams P{};
    } defer;
    bool seg_ready_pending = false;   // the compute stream has not been told yet to wait for the current set's tables (SegBuf::ready)
    bool host_tail = false;           // ICELK_HOST_TAIL=1: the tail of every detection through the host, as before round 4 (A/B)
    int tail_force_status = 0;        // ICELK_TAIL_FORCE_STATUS=1|2: the device verdict is forced to "host's tail" (tests of that path)
    long long tails_dev = 0, tails_host = 0;   // segments staged by the device-driven tail / by the host's
    bool use_order = true;                 // ICELK_NO_ORDER=1 launches in detector order (A/B measurements)
    // features this close to the frame border count as slow (launched first): from the window and pyramid depth of the
    // latest tracker call; ICELK_NO_BORDER_FIRST=1 turns the class off
    int border_px = (10 + kLkTileMargin + 2) << 2;
    bool border_first = true;
    bool pyr_per_level = false;            // ICELK_PYR_PER_LEVEL=1: one pyrDown launch per level (A/B, second statement)
    // pyramids built ahead (copy stream, beside a tracker launch) use one-wave workgroups, which fit into the slot of a
    // single retiring tracker wave (k_pyramid.hip); ICELK_PYR_AHEAD_WIDE=1 keeps the 256-thread geometry there too (A/B)
    bool pyr_ahead_one_wave = true;
    int fb_dist_form = ICELK_FB_HYPOT;     // icelk_set_fb_distance
    int lk_sum_mode = 0;                   // icelk_set_variant "lk_sums"
    int corner_variant = 0;                // icelk_set_variant "sobel_fma" (bits 0-1) | "eig_fma" (bit 2)
    int lk_kernel_flags = 0;               // icelk_set_lk_kernel: ICELK_FLAG_GENERIC_KERNEL / _ONE_PER_WAVE or 0
    // diagnostics: ICELK_LK_STAMPS=<file> records entry / exit time and placement of every workgroup of the LAST
    // segment tracker launch and writes them to the file when the handle is destroyed (tools/lk_stamps.py reads it)
    uint32_t* d_iters = nullptr;   // per-feature iteration counts of the latest tracker launch (while profiling is on)
    int iters_n = 0;
    unsigned long long* d_stamps = nullptr;
    size_t stamps_cap = 0;     // workgroups
    std::string stamps_path;
    unsigned long long* d_tracked = nullptr;   // 64 sharded counters
    unsigned long long* h_seg = nullptr;   // pinned: {alive tracks, features tracked}
    float *d_out_tracks = nullptr, *d_out_quality = nullptr;
    // projection epilogue: outputs x, y, u, v, speed (5 planes of proj_cap doubles) + keep bytes, grown on demand
    double* d_proj = nullptr;
    uint8_t* d_keep = nullptr;
    size_t proj_cap = 0;
    bool seg_active = false;
    bool seg_staged = false;   // the OTHER set holds a new segment waiting for icelk_seg_switch
    int staged_n = 0;

    int last_candidates = 0, last_accepted = 0;   // of the latest detection
    double prune_factor = 8.0;                    // candidates kept per corner wanted (top-K pruning, detect_begin)
    DetectJob job{};
    size_t reset_ncell = 0;    // the detector counters are known to be zero for grids up to this many cells
    bool counters_clean = false;
    int* h_counts = nullptr;   // pinned, device-visible: {candidates, accepted, undecided} of the job

    // profiling
    bool prof = false;
    bool prof_tracker_only = false;   // icelk_prof_enable(h, 2): every other kernel goes untimed (no event records on its stream)
    std::vector<ProfEvt> evts, evt_pool;
    int prof_launches[K_COUNT_] = {0};
    double prof_ms[K_COUNT_] = {0};
};

static std::string g_create_err;
static std::mutex g_mu;

#define HIPCHK(c, expr)                                                                      \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            (c)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                    \
            return ICELK_EHIP;                                                               \
        }                                                                                    \
    } while (0)

#define FAIL(c, code, msg)   \
    do {                     \
        (c)->err = (msg);    \
        return (code);       \
    } while (0)

static inline Ctx* C(icelk_t* h) { return reinterpret_cast<Ctx*>(h); }

static int check_launch(Ctx* c, const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        c->err = std::string(what) + ": " + hipGetErrorString(e);
        return ICELK_EHIP;
    }
    return ICELK_OK;
}

// ---- roctx ranges (tracing) --------------------------------------------------------------------
// ICELK_ROCTX=1: the host calls of the frame loop appear as named ranges in a rocprofv3 --marker-trace (tracker launch,
// detection begin / stage, candidates ahead, pyramid ahead, upload).  The marker library is looked up at run time
// (librocprofiler-sdk-roctx.so, else libroctx64.so): no link-time dependency, nothing is called when the variable is unset.
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        if (!getenv("ICELK_ROCTX")) return;
        void* lib = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) return;
        push = reinterpret_cast<int (*)(const char*)>(dlsym(lib, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
        if (!push || !pop) push = nullptr;
    }
};
static Roctx& roctx()
{
    static Roctx r;
    return r;
}
struct Range {
    bool on;
    explicit Range(const char* name) : on(roctx().push != nullptr)
    {
        if (on) roctx().push(name);
    }
    ~Range()
    {
        if (on) roctx().pop();
    }
};

struct ProfScope {
    Ctx* c;
    int id;
    hipStream_t st;
    ProfEvt ev{};
    ProfScope(Ctx* c_, int id_, hipStream_t st_ = nullptr) : c(c_), id(id_), st(st_ ? st_ : c_->stream)
    {
        on = c->prof && (!c->prof_tracker_only || id == K_LK_FB || id == K_LK_FB_PAIR || id == K_LK);
        if (on) {
            if (!c->evt_pool.empty()) {
                ev = c->evt_pool.back();
                c->evt_pool.pop_back();
            } else {
                hipEventCreate(&ev.a);
                hipEventCreate(&ev.b);
            }
            ev.id = id;
            hipEventRecord(ev.a, st);
        }
    }
    ~ProfScope()
    {
        if (on) {
            hipEventRecord(ev.b, st);
            c->evts.push_back(ev);
        }
    }
    bool on = false;
};

static void prof_drain(Ctx* c)
{
    for (auto& e : c->evts) {
        hipEventSynchronize(e.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
            c->prof_ms[e.id] += ms;
            c->prof_launches[e.id] += 1;
        }
        c->evt_pool.push_back(e);   // reused by later scopes, destroyed with the handle
    }
    c->evts.clear();
}

static int align_up(int v, int a) { return (v + a - 1) / a * a; }

// What the kernels that read and write whole dwords rely on (k_pyramid.hip stage 1 / copy_out, the tracker's tile loader,
// the corner kernels' staging): every level starts on a 256-B boundary, its row pitch is a multiple of 64 B >= the width
// rounded up to 4, so an aligned dword that STARTS inside a row (x = 0 mod 4, x < w) ends inside that row's pitch; and
// the allocation ends >= 256 B behind the last level, so a dword read that starts inside the last row of the last level
// stays inside the allocation.  Checked for every geometry a handle lays out (icelk_create, begin_frame).
static bool layout_ok(const Slot& s)
{
    for (int l = 0; l < kMaxLevels; l++) {
        const Level& L = s.lv[l];
        if (((uintptr_t)L.ptr & 255u) || (L.pitch % kPitchAlign) || L.pitch < ((L.w + 3) & ~3)) return false;
        if (L.ptr + (size_t)L.pitch * L.h + 256 > s.base + s.bytes) return false;
    }
    return true;
}

// level geometry of a w x h frame inside a slot allocation
static void layout_levels(Slot& s, int w, int h)
{
    size_t off = 0;
    int lw = w, lh = h;
    for (int l = 0; l < kMaxLevels; l++) {
        s.lv[l].w = lw;
        s.lv[l].h = lh;
        s.lv[l].pitch = align_up(lw, kPitchAlign);
        s.lv[l].ptr = s.base + off;
        off += (size_t)s.lv[l].pitch * lh;
        off = (off + 255) & ~(size_t)255;
        lw = (lw + 1) / 2;
        lh = (lh + 1) / 2;
    }
}

static size_t slot_bytes(int w, int h)
{
    size_t off = 0;
    int lw = w, lh = h;
    for (int l = 0; l < kMaxLevels; l++) {
        off += (size_t)align_up(lw, kPitchAlign) * lh;
        off = (off + 255) & ~(size_t)255;
        lw = (lw + 1) / 2;
        lh = (lh + 1) / 2;
    }
    return off + 256;
}

static int pyramid_top_level(int w, int h, int win_w, int win_h, int max_level)
{
    for (int level = 0; level <= max_level; level++) {
        w = (w + 1) / 2;
        h = (h + 1) / 2;
        if (w <= win_w || h <= win_h) return level;
    }
    return max_level;
}

static int check_slot(Ctx* c, int slot, bool need_image)
{
    if (slot < 0 || slot >= c->n_slots) FAIL(c, ICELK_EARG, "slot index out of range");
    if (need_image && c->slots[slot].levels_built < 1) FAIL(c, ICELK_ESTATE, "slot holds no frame");
    return ICELK_OK;
}

static int wait_event(Ctx* c, hipStream_t s, hipEvent_t e);

static int wait_slot(Ctx* c, int slot)
{
    Slot& s = c->slots[slot];
    if (s.pending) {
        if (int rcw = wait_event(c, c->stream, s.ready)) return rcw;
        s.pending = false;
    }
    return ICELK_OK;
}

static int mark_used(Ctx* c, int slot)
{
    Slot& s = c->slots[slot];
    HIPCHK(c, hipEventRecord(s.used_own, c->stream));
    s.used = s.used_own;
    return ICELK_OK;
}

// An event wait costs a barrier packet on the waiting queue, processed one after the other between its kernels: none
// when the event has completed already (also: was never recorded)
static int wait_event(Ctx* c, hipStream_t s, hipEvent_t e)
{
    if (hipEventQuery(e) == hipSuccess) return ICELK_OK;
    (void)hipGetLastError();   // hipErrorNotReady is the expected answer
    HIPCHK(c, hipStreamWaitEvent(s, e, 0));
    return ICELK_OK;
}

static int flush_deferred(Ctx* c);
static int flush_deferred_slot(Ctx* c, int slot);

static int begin_frame(Ctx* c, int slot, int w, int h)
{
    int rc = check_slot(c, slot, false);
    if (rc) return rc;
    if (w <= 0 || h <= 0) FAIL(c, ICELK_EARG, "empty image");
    if (w > c->max_w || h > c->max_h) FAIL(c, ICELK_ECAP, "frame larger than max_w x max_h of icelk_create");
    rc = flush_deferred_slot(c, slot);
    if (rc) return rc;
    Slot& s = c->slots[slot];
    // a detector launch on another stream may still read the frame this slot holds (compute-stream ingest paths
    // write level 0 right after this call; the copy-stream path waits for the same event itself)
    if (int rcw = wait_event(c, c->stream, s.det_used)) return rcw;
    if (int rcw = wait_event(c, c->stream, s.eig_used)) return rcw;
    // ... and a pyramid build enqueued ahead (icelk_build_pyramid_ahead) may still be writing levels >= 1 of the frame
    // this slot held: the ingest paths below clear `pending`, so the dependency is taken here
    if (s.pending) {
        if (int rcw = wait_event(c, c->stream, s.ready)) return rcw;
    }
    s.w = w;
    s.h = h;
    layout_levels(s, w, h);
    if (!layout_ok(s)) FAIL(c, ICELK_ECAP, "slot layout violates the dword-access invariant (internal)");
    s.levels_built = 0;
    s.gen++;
    return ICELK_OK;
}

// levels levels_built .. top_level of a slot on stream st: up to three levels per launch (k_pyramid.hip); the
// level-by-level kernel (k_image.hip) stays selectable (ICELK_PYR_PER_LEVEL=1) as the second statement of the arithmetic
static int build_levels(Ctx* c, Slot& s, int top_level, hipStream_t st)
{
    const bool per_level = c->pyr_per_level;
    while (s.levels_built < top_level + 1) {
        const int l = s.levels_built;
        const int n = per_level ? 1 : std::min(3, top_level + 1 - l);
        {
            ProfScope p(c, K_PYRDOWN, st);
            if (per_level) launch_pyrdown(st, s.lv[l - 1], s.lv[l]);
            else launch_pyramid_fused(st, s.lv, l - 1, n, c->pyr_ahead_one_wave && st == c->pyr_stream);
        }
        int rc = check_launch(c, "pyramid");
        if (rc) return rc;
        s.levels_built += n;
    }
    return ICELK_OK;
}

static int ensure_pyramid(Ctx* c, int slot, int top_level)
{
    Slot& s = c->slots[slot];
    int rc = wait_slot(c, slot);
    if (rc) return rc;
    if (top_level + 1 > kMaxLevels) FAIL(c, ICELK_EARG, "maxLevel too large");
    const bool build = s.levels_built < top_level + 1;
    rc = build_levels(c, s, top_level, c->stream);
    if (rc) return rc;
    return build ? mark_used(c, slot) : ICELK_OK;
}

static Pyramid pyramid_of(const Slot& s)
{
    Pyramid p;
    for (int l = 0; l < kMaxLevels; l++) p.lv[l] = s.lv[l];
    return p;
}

static int make_lk_params(Ctx* c, int w, int h, int win_w, int win_h, int max_level, int crit_type, int max_count,
                          double epsilon, int flags, double min_eig_thr, float fb_thr, LKParams* P)
{
    if (win_w <= 2 || win_h <= 2) FAIL(c, ICELK_EARG, "winSize must be > 2");
    if (max_level < 0) FAIL(c, ICELK_EARG, "maxLevel must be >= 0");
    if (max_level > kMaxLevels - 1) max_level = kMaxLevels - 1;
    if (win_w * win_h > 64 * 64) FAIL(c, ICELK_EARG, "winSize area above 4096 px is not supported");
    P->win_w = win_w;
    P->win_h = win_h;
    P->top_level = pyramid_top_level(w, h, win_w, win_h, max_level);
    if (!(crit_type & ICELK_CRIT_COUNT)) max_count = 30;
    else max_count = std::min(std::max(max_count, 0), 100);
    if (!(crit_type & ICELK_CRIT_EPS)) epsilon = 0.01;
    else epsilon = std::min(std::max(epsilon, 0.), 10.);
    P->max_count = max_count;
    P->eps2 = epsilon * epsilon;
    // A float evaluation of dx*dx + dy*dy is within 2^-22 (relative) of the double one OpenCV compares with eps^2;
    // outside a 2^-20 band around eps^2 it decides, inside the exact form runs (k_lk_multi.hip)
    if (P->eps2 < 1e-30) {
        P->eps2_lo = -1.f;
        P->eps2_hi = INFINITY;
    } else {
        P->eps2_lo = nextafterf((float)(P->eps2 * (1.0 - 1.0 / (1 << 20))), -INFINITY);
        P->eps2_hi = nextafterf((float)(P->eps2 * (1.0 + 1.0 / (1 << 20))), INFINITY);
    }
    P->flags = flags | c->lk_kernel_flags;
    P->min_eig_thr = (float)min_eig_thr;
    P->fb_thr = fb_thr;
    P->margin = 6;
    P->dist_form = c->fb_dist_form;
    P->sum_mode = c->lk_sum_mode;
    return ICELK_OK;
}

template <typename T>
static int dmalloc(Ctx* c, T** p, size_t count)
{
    hipError_t e = hipMalloc(reinterpret_cast<void**>(p), sizeof(T) * (count ? count : 1));
    if (e != hipSuccess) {
        c->err = std::string("hipMalloc: ") + hipGetErrorString(e);
        return ICELK_ENOMEM;
    }
    return ICELK_OK;
}

// the detection chain is ~20 short kernels; a high-priority queue keeps each of them from waiting behind
// the thousands of pending workgroups of the tracker launch it overlaps with
static hipError_t create_priority_stream(hipStream_t* s)
{
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (e != hipSuccess) return e;
    return hipStreamCreateWithPriority(s, hipStreamNonBlocking, greatest);
}

// ---- which hardware queue a side stream lands on matters --------------------------------------------------------------
// The runtime backs every HIP stream with a hardware queue, and the queues sit on a handful of dispatch pipes.  A pipe
// works on one dispatch at a time: while the tracker launch -- ten thousand workgroups, most of them waiting for a wave
// slot for most of the launch -- occupies its pipe, a kernel of ANOTHER queue on the same pipe is not even looked at
// until the last tracker workgroup has gone out.  Which pipe a new stream gets depends on how many queues the process
// has created before (the host framework's included), so it cannot be written down: it is measured.  Eight candidate
// high-priority streams are created; a probe fills the device from stream A with workgroups that idle for ~25 us each
// (~200 us in all) and times a one-wave kernel on stream B beside it -- it comes back after a few microseconds, or
// together with the filler.  The detection stream must not be held up by the compute stream; the candidates stream
// (one long kernel per detection) must not hold up the detection stream; the pyramid stream must be held up by neither
// the compute nor the candidates stream.  Measured on C2: 5 050 pairs/s with the three on pipes of their own, 4 000
// with the candidates stream behind the tracker's pipe.  ICELK_NO_STREAM_PROBE=1: creation order, no probe.
__global__ void k_probe_idle(unsigned ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
    for (int guard = 0; guard < 200000; guard++) {
        if (__builtin_amdgcn_s_memrealtime() - t0 >= ticks) break;
        __builtin_amdgcn_s_sleep(8);
    }
}

__global__ void k_probe_tick(unsigned* out)
{
    if (out) *out = 1u;
}

// fraction of the filler's duration (on `busy`) after which a one-wave kernel on `side` completed: ~0.1 when the two
// queues are served independently, ~1 when `side` waits for the filler's dispatch
static double probe_pair(hipStream_t busy, hipStream_t side, hipEvent_t e_busy, hipEvent_t e_side)
{
    using clk = std::chrono::steady_clock;
    hipStreamSynchronize(busy);
    hipStreamSynchronize(side);
    const auto t0 = clk::now();
    hipLaunchKernelGGL(k_probe_idle, dim3(65536), dim3(64), 0, busy, 2500u);
    hipEventRecord(e_busy, busy);
    hipLaunchKernelGGL(k_probe_tick, dim3(1), dim3(64), 0, side, (unsigned*)nullptr);
    hipEventRecord(e_side, side);
    hipEventSynchronize(e_side);
    const auto t1 = clk::now();
    hipEventSynchronize(e_busy);
    const auto t2 = clk::now();
    const double whole = std::chrono::duration<double>(t2 - t0).count();
    const double frac = whole > 0 ? std::chrono::duration<double>(t1 - t0).count() / whole : 1.0;
    if (getenv("ICELK_STREAM_PROBE_LOG"))
        fprintf(stderr, "icelk probe: busy %p side %p: side done after %.2f of %.0f us\n", (void*)busy, (void*)side, frac, 1e6 * whole);
    return frac;
}

static hipError_t create_side_streams(Ctx* c)
{
    if (getenv("ICELK_NO_STREAM_PROBE")) {
        hipError_t r = create_priority_stream(&c->det_stream);
        if (r == hipSuccess) r = create_priority_stream(&c->pyr_stream);
        if (r == hipSuccess) r = create_priority_stream(&c->eig_stream);
        if (r == hipSuccess) r = create_priority_stream(&c->tail_stream);
        return r;
    }
    constexpr int NC = 8;
    hipStream_t cand[NC] = {nullptr};
    hipEvent_t ea = nullptr, eb = nullptr;
    hipError_t r = hipEventCreateWithFlags(&ea, hipEventDisableTiming);
    if (r == hipSuccess) r = hipEventCreateWithFlags(&eb, hipEventDisableTiming);
    for (int i = 0; i < NC && r == hipSuccess; i++) r = create_priority_stream(&cand[i]);
    if (r == hipSuccess) {
        probe_pair(c->own_stream, cand[0], ea, eb);   // code object load, clocks
        // every candidate beside the compute stream, twice; "held up" = clearly later than the quickest one
        double beside[NC], quickest = 1.0;
        for (int i = 0; i < NC; i++) {
            beside[i] = std::min(probe_pair(c->own_stream, cand[i], ea, eb), probe_pair(c->own_stream, cand[i], ea, eb));
            quickest = std::min(quickest, beside[i]);
        }
        const double limit = std::max(1.6 * quickest, quickest + 0.12);
        bool used[NC] = {false};
        auto pick = [&](auto ok) {
            for (int i = 0; i < NC; i++)
                if (!used[i] && beside[i] <= limit && ok(cand[i])) { used[i] = true; return i; }
            for (int i = 0; i < NC; i++)        // nothing passes (fewer pipes than assumed): the least held up of the rest
                if (!used[i]) { used[i] = true; return i; }
            return 0;
        };
        const int d = pick([&](hipStream_t) { return true; });
        c->det_stream = cand[d];
        const int e = pick([&](hipStream_t s) { return probe_pair(s, c->det_stream, ea, eb) <= limit; });
        c->eig_stream = cand[e];
        const int q = pick([&](hipStream_t s) { return probe_pair(c->eig_stream, s, ea, eb) <= limit; });
        c->pyr_stream = cand[q];
        const int t = pick([&](hipStream_t s) { return probe_pair(c->eig_stream, s, ea, eb) <= limit; });
        c->tail_stream = cand[t];
        c->side_pick[0] = d;
        c->side_pick[1] = e;
        c->side_pick[2] = q;
        c->side_pick[3] = t;
        c->probe_limit = limit;
        c->probe_quickest = quickest;
        if (getenv("ICELK_STREAM_PROBE_LOG"))
            fprintf(stderr, "icelk probe: detection = candidate %d, candidates stream = %d, pyramid = %d (limit %.2f)\n", d, e, q, limit);
        for (int i = 0; i < NC; i++)
            if (!used[i]) hipStreamDestroy(cand[i]);
    } else {
        for (auto s : cand)
            if (s) hipStreamDestroy(s);
    }
    if (ea) hipEventDestroy(ea);
    if (eb) hipEventDestroy(eb);
    return r;
}

static void det_save(Ctx* c);

static void destroy_ctx(Ctx* c)
{
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    prof_drain(c);
    for (auto& e : c->evt_pool) {
        hipEventDestroy(e.a);
        hipEventDestroy(e.b);
    }
    if (c->d_stamps) {
        std::vector<unsigned long long> hs(3 * c->stamps_cap);
        if (hipMemcpy(hs.data(), c->d_stamps, hs.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            if (FILE* f = fopen(c->stamps_path.c_str(), "wb")) {
                fwrite(hs.data(), 8, hs.size(), f);
                fclose(f);
            }
        }
        hipFree(c->d_stamps);
    }
    if (c->d_iters) hipFree(c->d_iters);
    for (void* b : c->tmpl.buf)
        if (b) hipFree(b);
    for (auto& s : c->slots) {
        if (s.base) hipFree(s.base);
        if (s.ready) hipEventDestroy(s.ready);
        if (s.frame_ev) hipEventDestroy(s.frame_ev);
        if (s.used_own) hipEventDestroy(s.used_own);
        if (s.det_used) hipEventDestroy(s.det_used);
        if (s.eig_used) hipEventDestroy(s.eig_used);
    }
    if (c->det_stream) hipStreamSynchronize(c->det_stream);
    if (c->eig_stream) hipStreamSynchronize(c->eig_stream);
    if (c->det_done) hipEventDestroy(c->det_done);
    if (c->corners_free) hipEventDestroy(c->corners_free);
    if (c->det_stream) hipStreamDestroy(c->det_stream);
    if (c->eig_stream) hipStreamDestroy(c->eig_stream);
    for (auto& b : c->sb) {
        if (b.used_own) hipEventDestroy(b.used_own);
        if (b.ready) hipEventDestroy(b.ready);
    }
    for (auto& e : c->launch_ev)
        if (e) hipEventDestroy(e);
    for (auto& e : c->eo)
        if (e.done) hipEventDestroy(e.done);
    det_save(c);
    for (auto& S : c->dset) {
        if (S.h_counts) hipHostFree(S.h_counts);
        if (S.counts_ev) hipEventDestroy(S.counts_ev);
        if (S.tail_done) hipEventDestroy(S.tail_done);
    }
    {
        DetectScratch& E = c->dset[c->dset_cur ^ 1].D;     // the other set's own arrays (the working copy's are freed below)
        void* ep[] = {E.cand, E.cand_count, E.cell_count, E.cell_start, E.cell_fill, E.chunk_tot, E.cell_cand, E.state, E.undecided,
                      E.acc, E.acc_sorted, E.acc_count, E.key_hist, E.prune_key, E.sort_tmp, E.tail_ctl, E.tail_resp, E.tail_bins};
        for (void* q : ep)
            if (q) hipFree(q);
    }
    for (auto& e : c->eo) {
        void* fp[] = {e.acand, e.acount, e.amaxc, e.amaxn, e.aemax, e.fmax_key, e.aties};
        for (void* q : fp)
            if (q) hipFree(q);
    }
    for (auto& e : c->eo) {   // [0] and [1] are in the list below
        if (&e == &c->eo[2]) {
            if (e.raw) hipFree(e.raw);
            if (e.blk_count) hipFree(e.blk_count);
            if (e.max_key) hipFree(e.max_key);
        }
    }
    if (c->h_seg) hipHostFree(c->h_seg);
    void* ptrs[] = {c->d_bgr, c->d_mask, c->d_p0, c->d_p1, c->d_p0r, c->d_err_f, c->d_err_b, c->d_dist, c->d_corners,
                    c->d_st_f, c->d_st_b, c->d_valid, c->D.eig, c->D.cand, c->D.cand_count,
                    c->D.cell_count, c->D.cell_start, c->D.cell_fill, c->D.chunk_tot, c->D.cell_cand, c->D.state, c->D.undecided,
                    c->D.acc, c->D.acc_sorted, c->D.acc_count, c->eo[0].raw, c->eo[1].raw, c->eo[0].blk_count, c->eo[1].blk_count,
                    c->eo[0].max_key, c->eo[1].max_key, c->D.key_hist, c->D.prune_key, c->D.sort_tmp, c->D.tail_ctl,
                    c->D.tail_resp, c->D.tail_bins, c->d_tracked,
                    c->d_out_tracks, c->d_out_quality, c->d_proj, c->d_keep};
    for (void* p : ptrs)
        if (p) hipFree(p);
    for (auto& S : c->sb) {
        void* sp[] = {S.live, S.alive, S.order, S.order_border, S.tracks, S.quality};
        for (void* p : sp)
            if (p) hipFree(p);
    }
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    for (auto q : c->copy_hi)
        if (q) {
            hipStreamSynchronize(q);
            hipStreamDestroy(q);
        }
    if (c->copy_stream) hipStreamDestroy(c->copy_stream);
    if (c->copy_stream2) {
        hipStreamSynchronize(c->copy_stream2);
        hipStreamDestroy(c->copy_stream2);
    }
    for (auto q : c->copy_more)
        if (q) {
            hipStreamSynchronize(q);
            hipStreamDestroy(q);
        }
    if (c->pyr_stream) {
        hipStreamSynchronize(c->pyr_stream);
        hipStreamDestroy(c->pyr_stream);
    }
    if (c->tail_stream) {
        hipStreamSynchronize(c->tail_stream);
        hipStreamDestroy(c->tail_stream);
    }
    delete c;
}

// ---- detector core shared by icelk_good_features and icelk_seg_detect --------------------------
// detect_begin enqueues K6..K8 on the detection stream and returns at once; detect_finish waits for the
// counts, sorts the accepted corners and leaves the first *n_out of them in c->d_corners (device), in
// response order.
constexpr int kCountsSeq = 8;   // word of the pinned counts that carries the sequence number of the publication
__global__ void k_publish_counts(const int* __restrict__ cand, const int* __restrict__ acc,
                                 const int* __restrict__ undecided, const unsigned* __restrict__ prune_key,
                                 int* __restrict__ host_out, int seq)
{
    host_out[0] = *cand;
    host_out[1] = *acc;
    host_out[2] = *undecided;
    host_out[3] = (int)(*prune_key != 0u);
    __threadfence_system();
    // the host polls this word (fetch_counts): it learns of the counts when they land in its memory, not when the
    // runtime has processed the completion signal of an event behind this kernel and woken the waiting thread
    __hip_atomic_store(host_out + kCountsSeq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static void det_save(Ctx* c)
{
    Ctx::DetSet& S = c->dset[c->dset_cur];
    S.D = c->D;
    S.job = c->job;
    S.h_counts = c->h_counts;
    S.counts_seq = c->counts_seq;
    S.counts_ev = c->counts_ev;
    S.tail_done = c->tail_done;
    S.reset_ncell = c->reset_ncell;
    S.counters_clean = c->counters_clean;
    S.eo_active = c->eo_active;
}

static void det_load(Ctx* c, int k)
{
    det_save(c);
    const Ctx::DetSet& S = c->dset[k];
    c->D = S.D;
    c->job = S.job;
    c->h_counts = S.h_counts;
    c->counts_seq = S.counts_seq;
    c->counts_ev = S.counts_ev;
    c->tail_done = S.tail_done;
    c->reset_ncell = S.reset_ncell;
    c->counters_clean = S.counters_clean;
    c->eo_active = S.eo_active;
    c->dset_cur = k;
}

// the set of the oldest detection in flight, or -1
static int det_oldest(Ctx* c)
{
    det_save(c);
    int k = -1;
    for (int i = 0; i < 2; i++)
        if (c->dset[i].job.active && (k < 0 || c->dset[i].job.seq < c->dset[k].job.seq)) k = i;
    return k;
}

// a set with no detection in flight, or -1
static int det_free(Ctx* c)
{
    det_save(c);
    for (int i = 0; i < 2; i++)
        if (!c->dset[i].job.active) return i;
    return -1;
}

// a candidate buffer no detection in flight reads: the one that holds what `want` asks for if there is one, else one
// without valid content, else any
static int eo_free(Ctx* c, const Ctx::EigOut* want)
{
    det_save(c);
    bool busy[3] = {false, false, false};
    for (int i = 0; i < 2; i++)
        if (c->dset[i].job.active) busy[c->dset[i].eo_active] = true;
    int pick = -1;
    for (int i = 0; i < 3; i++) {
        if (busy[i]) continue;
        const Ctx::EigOut& e = c->eo[i];
        if (want && e.valid && e.slot == want->slot && e.gen == want->gen && e.block_size == want->block_size &&
            e.use_mask == want->use_mask && e.mask_gen == want->mask_gen)
            return i;
        if (pick < 0 || (c->eo[pick].valid && !e.valid)) pick = i;
    }
    return pick;
}

// The counts of the working set's detection go to its pinned host words behind whatever is queued on the detection
// stream, and an event of the set marks them.  detect_begin ends with this, so the host round trip of that detection
// waits for ITS kernels only -- not for the min-distance stage of the next detection that may be queued behind them.
static int publish_counts(Ctx* c)
{
    const DetectJob& J = c->job;
    c->counts_seq = (c->counts_seq + 1) & 0x3fffffff;
    hipLaunchKernelGGL(k_publish_counts, dim3(1), dim3(1), 0, c->det_stream, J.cand_count_ptr, c->D.acc_count,
                       c->D.undecided + suppress_launch_count() - 1, c->D.prune_key, c->h_counts, c->counts_seq);
    HIPCHK(c, hipEventRecord(c->counts_ev, c->det_stream));
    return ICELK_OK;
}

// have the counts published last arrived?  (the sequence word, no runtime call)
static inline bool counts_here(const int* h_counts, int seq)
{
    return __atomic_load_n(h_counts + kCountsSeq, __ATOMIC_ACQUIRE) == seq;
}

// Waits for the counts by polling the pinned sequence word; the event is looked at now and then, so that a failed
// kernel ends the wait with its error instead of hanging it (ICELK_EVENT_WAIT=1: hipEventSynchronize, as before).
static int fetch_counts(Ctx* c, bool published = false)
{
    if (!published) {
        int rc = publish_counts(c);
        if (rc) return rc;
    }
    static const bool by_event = getenv("ICELK_EVENT_WAIT") != nullptr;
    if (!by_event) {
        for (unsigned it = 1;; it++) {
            if (counts_here(c->h_counts, c->counts_seq)) return ICELK_OK;
            if ((it & 4095u) == 0) {
                const hipError_t q = hipEventQuery(c->counts_ev);
                if (q == hipSuccess) break;             // complete: the synchronize below returns at once
                if (q != hipErrorNotReady) HIPCHK(c, q);
                (void)hipGetLastError();
            }
            __builtin_ia32_pause();
        }
    }
    HIPCHK(c, hipEventSynchronize(c->counts_ev));
    return ICELK_OK;
}

// make eo[idx] the buffer the detector stages read (and the non-prepared corner kernel writes)
static void activate_eig_out(Ctx* c, int idx)
{
    c->eo_active = idx;
    const Ctx::EigOut& e = c->eo[idx];
    c->D.raw = e.raw;
    c->D.blk_count = e.blk_count;
    c->D.max_key = e.max_key;
    c->D.acand = e.acand;
    c->D.acount = e.acount;
    c->D.amaxc = e.amaxc;
    c->D.amaxn = e.amaxn;
    c->D.aemax = e.aemax;
    c->D.fmax_key = e.fmax_key;
    c->D.aties = e.aties;
    c->D.src_nblk = e.nblk;
    c->D.src_region = e.region;
}

// Corner candidates of a frame ahead of its detection (fused kernel only; anything else is left to
// detect_begin).  Runs on eig_stream into the spare buffer; detect_begin adopts it when slot, frame
// generation, blockSize and mask still match.
static int detect_prepare(Ctx* c, int slot, int use_mask, int block_size)
{
    Range rg("icelk detect_prepare (corner candidates ahead)");
    int rc = check_slot(c, slot, true);
    if (rc) return rc;
    if (!fused_block_size(block_size) || getenv("ICELK_GENERIC_CORNERS") || c->corner_variant) return ICELK_OK;
    Slot& s = c->slots[slot];
    const uint8_t* mask = nullptr;
    if (use_mask) {
        if (!c->has_mask) FAIL(c, ICELK_ESTATE, "use_mask set but no mask uploaded");
        if (c->mask_w != s.w || c->mask_h != s.h) FAIL(c, ICELK_EARG, "mask size differs from the frame");
        mask = c->d_mask;
    }
    Ctx::EigOut want;
    want.slot = slot;
    want.gen = s.gen;
    want.block_size = block_size;
    want.use_mask = use_mask;
    want.mask_gen = c->mask_gen;
    Ctx::EigOut& e = c->eo[eo_free(c, &want)];
    if (e.valid && e.slot == slot && e.gen == s.gen && e.block_size == block_size && e.use_mask == use_mask &&
        e.mask_gen == c->mask_gen)
        return ICELK_OK;   // already there
    const hipStream_t es = c->eig_stream;
    // level 0 only: `ready` would also wait for a pyramid built ahead, which the detector never reads
    if (int rcw = wait_event(c, es, s.frame_ev)) return rcw;
    HIPCHK(c, hipMemsetAsync(e.max_key, 0, sizeof(unsigned), es));
    DetectScratch T = c->D;
    T.raw = e.raw;
    T.blk_count = e.blk_count;
    T.max_key = e.max_key;
    T.acand = e.acand;
    T.acount = e.acount;
    T.amaxc = e.amaxc;
    T.amaxn = e.amaxn;
    T.aemax = e.aemax;
    T.fmax_key = e.fmax_key;
    T.aties = e.aties;
    // the quality level is not known yet: the one of the latest detection begun on this handle is taken (0 at first: every
    // local maximum gets its exact key); detect_begin adopts the result only if its own level is not lower
    const double prep_quality = c->prep_quality;
    {
        ProfScope p(c, K_EIG, es);
        launch_candidates(es, T, s.lv[0], block_size, mask, c->mask_pitch, prep_quality, false, nullptr, 0, true);
    }
    rc = check_launch(c, "corner candidates (prepared)");
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(e.done, es));
    HIPCHK(c, hipEventRecord(s.eig_used, es));
    e.nblk = T.src_nblk;
    e.region = T.src_region;
    e.valid = true;
    e.slot = slot;
    e.gen = s.gen;
    e.block_size = block_size;
    e.use_mask = use_mask;
    e.mask_gen = c->mask_gen;
    e.quality = prep_quality;
    return ICELK_OK;
}

// for_segment: the corners start a segment (icelk_seg_detect_begin) -- the tail of the detection is then enqueued here, behind
// the min-distance stage, driven by the device-side counts (k_tail.hip), and writes the segment's tables into the set it
// reserves; otherwise (icelk_good_features) detect_finish runs the tail after the host round trip
static int detect_begin(Ctx* c, int slot, int use_mask, int max_corners, double quality, double min_distance,
                        int block_size, bool for_segment)
{
    Range rg("icelk detect_begin (candidates + min-distance stage)");
    int rc = check_slot(c, slot, true);
    if (rc) return rc;
    if (!(quality > 0) || min_distance < 0 || block_size <= 0) FAIL(c, ICELK_EARG, "bad detector parameters");
    if (min_eig_lds_bytes(block_size) > 150 * 1024) FAIL(c, ICELK_EARG, "blockSize too large");
    {
        const int k = det_free(c);
        if (k < 0) FAIL(c, ICELK_ESTATE, "two detections are in flight already");
        det_load(c, k);
    }
    Slot& s = c->slots[slot];
    const hipStream_t ds = c->det_stream;
    const int w = s.w, h = s.h;
    const uint8_t* mask = nullptr;
    if (use_mask) {
        if (!c->has_mask) FAIL(c, ICELK_ESTATE, "use_mask set but no mask uploaded");
        if (c->mask_w != w || c->mask_h != h) FAIL(c, ICELK_EARG, "mask size differs from the frame");
        mask = c->d_mask;
    }
    DetectScratch& D = c->D;
    size_t ncell = 0;
    if (min_distance >= 1) {
        const int cell = (int)lrint(min_distance);
        ncell = (size_t)((w + cell - 1) / cell) * ((h + cell - 1) / cell);
        if (ncell + 1 > c->ncell_cap) FAIL(c, ICELK_ECAP, "cell grid larger than allocated");
    }
    // the frame must be in the slot (ingest on the compute or the copy stream) and the previous corner list
    // must have been consumed before this detection overwrites it; nothing else orders the two streams
    // the frame must be in the slot, and the tail of this set's previous detection (it sorted this set's accepted keys and
    // reset its counters on the tail stream) must be through
    if (int rcw = wait_event(c, ds, s.frame_ev)) return rcw;   // level 0 only (see detect_prepare)
    if (int rcw = wait_event(c, ds, c->tail_done)) return rcw;
    const bool generic = getenv("ICELK_GENERIC_CORNERS") != nullptr || c->corner_variant != 0;
    // counters are normally left zeroed by the previous detection (the reset runs after its last kernel,
    // off the critical path); reset here only the first time or when the cell grid grew
    const bool need_reset = !c->counters_clean || ncell > c->reset_ncell;
    Ctx::EigOut want;
    want.slot = slot;
    want.gen = s.gen;
    want.block_size = block_size;
    want.use_mask = use_mask;
    want.mask_gen = c->mask_gen;
    const int spare_idx = eo_free(c, &want);
    Ctx::EigOut& spare = c->eo[spare_idx];
    const bool prepared = spare.valid && spare.slot == slot && spare.gen == s.gen && spare.block_size == block_size &&
                          spare.use_mask == use_mask && spare.mask_gen == c->mask_gen && !generic && spare.quality <= quality;
    c->prep_quality = quality;
    spare.valid = false;   // adopted below, or overwritten: either way it is not offered again
    if (prepared) {
        if (need_reset) launch_detect_reset(ds, D, (int)ncell, 1);
        activate_eig_out(c, spare_idx);
        if (int rcw = wait_event(c, ds, spare.done)) return rcw;
    } else {
        ProfScope p(c, K_EIG, ds);
        // this detection's candidates go into a buffer no detection in flight reads; a prepare launch that wrote it
        // (for another frame) must be through, and its maximum starts from zero
        activate_eig_out(c, spare_idx);
        if (int rcw = wait_event(c, ds, spare.done)) return rcw;
        HIPCHK(c, hipMemsetAsync(spare.max_key, 0, sizeof(unsigned), ds));
        if (need_reset) launch_detect_reset(ds, D, (int)ncell, 1);
        launch_candidates(ds, D, s.lv[0], block_size, mask, c->mask_pitch, quality, generic, nullptr, c->corner_variant);
        HIPCHK(c, hipEventRecord(s.det_used, ds));   // nothing after this launch reads the frame
    }
    c->counters_clean = false;
    rc = check_launch(c, "corner candidates");
    if (rc) return rc;
    DetectJob& J = c->job;
    J.w = w;
    J.h = h;
    J.quality = quality;
    J.min_distance = min_distance;
    J.ncell = ncell;
    // top-K pruning (k_corners.hip): worthwhile when maxCorners is a real cap.  Only the strongest candidates can be among
    // the first maxCorners accepted ones; how many to keep follows the share that survived the minDistance rule in the
    // detection before (with half as many again; 8x to begin with and after a shortfall), and detect_finish verifies
    // that maxCorners corners came out -- else the stage runs once more on all candidates
    J.prune_want = 0;
    if (min_distance >= 1 && max_corners > 0 && max_corners <= (1 << 24) && !getenv("ICELK_NO_PRUNE"))
        J.prune_want = (int)std::min(8.0 * max_corners, std::ceil(c->prune_factor * max_corners));
    J.dev_tail = false;
    J.seg_set = -1;
    J.max_corners = max_corners;
    const bool dev_tail = for_segment && !c->host_tail && min_distance >= 1;
    if (min_distance >= 1) {
        J.cand_count_ptr = D.cell_start + ncell;
        ProfScope p(c, K_SUPPRESS, ds);
        launch_min_distance(ds, D, w, h, min_distance, quality, J.prune_want, dev_tail);
        if (dev_tail)
            launch_tail_gather(ds, D, (int)ncell, quality, suppress_launch_count() - 1, max_corners, c->max_pts, c->tail_force_status);
    } else {
        J.cand_count_ptr = D.cand_count;
        launch_flatten(ds, D, quality);
    }
    rc = check_launch(c, "min_distance");
    if (rc) return rc;
    if (dev_tail) {
        // the set the new segment goes into: behind the current one, the staged one and the one the other detection in
        // flight has reserved (segments are staged and switched to in the order their detections were begun)
        const Ctx::DetSet& other = c->dset[c->dset_cur ^ 1];
        const int ahead = (c->seg_staged ? 1 : 0) + (other.job.active ? 1 : 0);
        const int target = (c->sb_cur + 1 + ahead) % kSegSets;
        Ctx::SegBuf& nb = c->sb[target];
        // launches that still touch that set (a segment closed several switches ago) must be through
        if (int rcw = wait_event(c, ds, nb.used)) return rcw;
        c->counts_seq = (c->counts_seq + 1) & 0x3fffffff;
        {
            ProfScope p(c, K_EMIT, ds);
            launch_tail_device(ds, D, quality, nb.live, nb.alive, nb.tracks, kMaxVert, c->use_order ? nb.order : nullptr,
                               nb.order_border, tail_order_geometry(w, h, c->border_px), tail_reset_of(D, (int)ncell),
                               c->h_counts, kCountsSeq, c->counts_seq);
        }
        rc = check_launch(c, "tail (device-driven)");
        if (rc) return rc;
        HIPCHK(c, hipEventRecord(nb.ready, ds));
        HIPCHK(c, hipEventRecord(c->counts_ev, ds));
        J.dev_tail = true;
        J.seg_set = target;
    } else {
        rc = publish_counts(c);
        if (rc) return rc;
    }
    J.active = true;
    J.seq = ++c->job_seq;
    return ICELK_OK;
}

// seg != null (seg_stage): the corners start a segment in that set -- corner list, the segment's tables and the counter
// reset go out as ONE launch (k_tail) instead of three; *seg_done tells seg_stage that the tables are written
// *dev_done: the device-driven tail has written everything (tables AND launch order): nothing is left to enqueue
static int detect_finish(Ctx* c, int max_corners, int cap, int* n_out, Ctx::SegBuf* seg = nullptr, bool* seg_done = nullptr,
                         bool* dev_done = nullptr)
{
    Range rg("icelk detect_finish (host round trip, sort, corner list)");
    if (seg_done) *seg_done = false;
    if (dev_done) *dev_done = false;
    {
        const int k = det_oldest(c);
        if (k < 0) FAIL(c, ICELK_ESTATE, "no detection in flight");
        det_load(c, k);
    }
    DetectJob& J = c->job;
    J.active = false;
    *n_out = 0;
    const hipStream_t ds = c->det_stream;
    // Everything behind the host round trip -- sort, corner list, the reset of this set's counters, and in seg_stage the
    // new segment's tables -- goes to the tail stream: the detection stream may already hold the min-distance stage of
    // the NEXT detection (the other set), and the two have nothing in common but d_corners, which only tails touch.
    const hipStream_t ts = c->tail_stream;
    DetectScratch& D = c->D;
    int rc = fetch_counts(c, true);   // the one host round trip of a detection: {candidates, accepted, undecided}
    if (rc) return rc;
    if (J.dev_tail) {
        // the tail ran on the device already; the host adopts its verdict
        const int status = c->h_counts[5];
        if (!seg || seg != &c->sb[J.seg_set] || max_corners != J.max_corners)
            FAIL(c, ICELK_ESTATE, "the segment staged is not the one its detection was begun for (set / maxCorners differ)");
        if (status == TAIL_OVERFLOW) FAIL(c, ICELK_ECAP, "more corners than the output capacity (raise max_pts)");
        if (status == TAIL_OK) {
            const int total = c->h_counts[1], n = c->h_counts[4];
            if (n > cap) FAIL(c, ICELK_ECAP, "more corners than the output capacity (raise max_pts)");
            if (J.prune_want > 0 && max_corners > 0)
                c->prune_factor = !c->h_counts[3] || total <= 0 ? 8.0 : std::min(8.0, std::max(2.0, 1.5 * (double)c->h_counts[0] / total));
            c->last_candidates = c->h_counts[0];
            c->last_accepted = total;
            c->reset_ncell = J.ncell;
            c->counters_clean = true;    // k_tail_order left them zeroed
            c->tails_dev++;
            *n_out = n;
            if (seg_done) *seg_done = true;
            if (dev_done) *dev_done = true;
            return ICELK_OK;
        }
        // not converged / pruned set fell short: the host's tail below, on the counts the device published
    }
    c->tails_host++;
    const unsigned long long* sorted = nullptr;
    int total = 0;
    if (J.min_distance >= 1) {
        auto converge = [&]() -> int {
            for (int guard = 0; c->h_counts[2] != 0; guard++) {
                if (guard > 100000) FAIL(c, ICELK_EHIP, "min-distance suppression did not converge");
                continue_min_distance(ds, D, J.w, J.h, J.min_distance);
                int r = fetch_counts(c);
                if (r) return r;
            }
            return ICELK_OK;
        };
        if ((rc = converge())) return rc;
        bool redone = false;
        if (J.prune_want > 0 && c->h_counts[3] && (max_corners <= 0 || c->h_counts[1] < max_corners)) {
            // the pruned candidate set did not yield maxCorners corners: redo the stage on all candidates
            redone = true;
            launch_detect_reset(ds, D, (int)J.ncell, 0);
            launch_min_distance(ds, D, J.w, J.h, J.min_distance, J.quality, 0);
            if ((rc = check_launch(c, "min_distance (unpruned)"))) return rc;
            if ((rc = fetch_counts(c))) return rc;
            if ((rc = converge())) return rc;
        }
        total = c->h_counts[1];
        if (J.prune_want > 0 && max_corners > 0) {
            // next time: candidates per accepted corner as seen now, and half as many again; a detection that fell short
            // (it was redone above) or was not pruned at all starts over at 8x
            const bool fell_short = !c->h_counts[3] || redone;
            c->prune_factor = fell_short || total <= 0 ? 8.0 : std::min(8.0, std::max(2.0, 1.5 * (double)c->h_counts[0] / total));
        }
        c->last_candidates = c->h_counts[0];
        c->last_accepted = total;
        if (total == 0) return ICELK_OK;
        sort_keys_desc(ts, D, D.acc, D.acc_sorted, total);
        sorted = D.acc_sorted;
    } else {
        total = c->h_counts[0];
        c->last_candidates = total;
        c->last_accepted = total;
        if (total == 0) return ICELK_OK;
        sort_keys_desc(ts, D, D.cand, D.cell_cand, total);
        sorted = D.cell_cand;
    }
    rc = check_launch(c, "sort");
    if (rc) return rc;
    int n = total;
    if (max_corners > 0 && n > max_corners) n = max_corners;
    if (n > cap || n > c->max_pts) FAIL(c, ICELK_ECAP, "more corners than the output capacity (raise max_pts)");
    static const bool split_tail = getenv("ICELK_SPLIT_TAIL") != nullptr;   // A/B: the three launches of before
    if (seg && !split_tail) {
        // launches that still touch the segment set (a segment closed two switches ago) must be through
        if (int rcw = wait_event(c, ts, seg->used)) return rcw;
        {
            ProfScope p(c, K_EMIT, ts);
            launch_tail_fused(ts, sorted, n, c->d_corners, seg->live, seg->alive, seg->tracks, kMaxVert, D, (int)J.ncell, 1);
        }
        rc = check_launch(c, "tail");
        if (rc) return rc;
        HIPCHK(c, hipEventRecord(c->det_done, ts));
        HIPCHK(c, hipEventRecord(c->tail_done, ts));
        *seg_done = true;
    } else {
        {
            ProfScope p(c, K_EMIT, ts);
            launch_emit_corners(ts, sorted, n, J.w, c->d_corners);
        }
        rc = check_launch(c, "emit");
        if (rc) return rc;
        HIPCHK(c, hipEventRecord(c->det_done, ts));
        launch_detect_reset(ts, D, (int)J.ncell, 1);   // for this set's next detection, which waits for tail_done
        HIPCHK(c, hipEventRecord(c->tail_done, ts));
    }
    c->reset_ncell = J.ncell;
    c->counters_clean = true;
    *n_out = n;
    return ICELK_OK;
}

static int detect_core(Ctx* c, int slot, int use_mask, int max_corners, double quality, double min_distance,
                       int block_size, int cap, int* n_out)
{
    *n_out = 0;
    // begin + finish in one go: the detection finished must be the one begun here
    if (det_oldest(c) >= 0) FAIL(c, ICELK_ESTATE, "a detection is in flight (icelk_seg_detect_begin without _stage / _finish)");
    int rc = detect_begin(c, slot, use_mask, max_corners, quality, min_distance, block_size, false);
    if (rc) return rc;
    return detect_finish(c, max_corners, cap, n_out);
}

// the compute stream must see the current segment set initialised (detection stream) before it touches it
static int seg_wait(Ctx* c)
{
    if (c->seg_ready_pending) {
        if (int rcw = wait_event(c, c->stream, c->sb[c->sb_cur].ready)) return rcw;
        c->seg_ready_pending = false;
    }
    return ICELK_OK;
}

// the segment-pair job of set `set` across slots s0 -> s1; `primary` jobs also fill the handle's per-feature diagnostic
// arrays (positions, status, error, distance of the latest launch) -- one job per launch can own them
static LKJob seg_job(Ctx* c, int set, const Slot& s0, const Slot& s1, const LKParams& P, bool primary)
{
    Ctx::SegBuf& S = c->sb[set];
    LKJob j{};
    j.I = pyramid_of(s0);
    j.J = pyramid_of(s1);
    j.n = S.upper;
    LKBuffers& B = j.B;
    B.p_in = S.live;
    if (primary) {
        B.p_fwd = c->d_p1;
        B.st_fwd = c->d_st_f;
        B.p_bwd = c->d_p0r;
        B.st_bwd = c->d_st_b;
        // no err_fwd / err_bwd: the segment loop has no use for the residual error (s1:323,326 drop it), and the tracker
        // kernels skip forming it when nobody takes it
        B.dist = c->d_dist;
        B.valid = c->d_valid;
    }
    B.seg_alive = S.alive;
    B.order = c->use_order ? S.order : nullptr;
    B.order_border = c->use_order ? S.order_border : nullptr;
    // Dealing the sorted sequence to the XCDs pays while neighbouring windows barely overlap (C2: 244 -> 233 us);
    // with dense features every XCD would work on one spot of the frame at a time and its L2 channels
    // serialise (REF: 2 060 us walking the table linearly, 2 680 us dealt, 2 230 us unsorted)
    const double overlap = (double)(P.win_w + 12) * (P.win_h + 12) * S.upper / ((double)s0.w * s0.h);
    B.order_plain = overlap >= 2.0 ? 1 : 0;
    B.seg_xy = S.live;
    B.seg_tracks = S.tracks;
    B.seg_quality = S.quality;
    B.seg_vert = S.vert;
    B.seg_max_vert = kMaxVert;
    B.seg_tracked = c->d_tracked;
    // templates: taken from the pair before if it left them for this vertex, left for the pair after unless this is the
    // segment's last
    const int quads = lk_fast_eligible(P) ? lk_template_quads(P.win_w, P.win_h) : 0;
    const int key = (P.win_w << 16) | (P.win_h << 8) | (P.top_level + 1);
    // (the templates were built on the frame and pyramid that sat in the pair's second slot: they serve the next pair
    // only if that very frame is now its first)
    const int slot0 = (int)(&s0 - c->slots.data()), slot1 = (int)(&s1 - c->slots.data());
    bool take = quads > 0 && S.tmpl_for == S.vert && S.tmpl_key == key && S.tmpl_slot == slot0 && S.tmpl_gen == s0.gen;
    S.tmpl_for = -1;
    if (quads > 0 && !c->tmpl.off) {
        const bool last = c->track_len_hint > 0 && S.vert >= c->track_len_hint;
        const size_t per_row = (size_t)(P.top_level + 1) * ((size_t)quads * 64) * 16;
        const size_t need = (size_t)std::max(S.upper, 1) * per_row;
        // The two tables are sized ONCE per row geometry (window, levels), for max_pts rows within the handle's template
        // budget (ICELK_TEMPLATE_BUDGET_MB, default 8 GB for both; never more than half of what the device has free) --
        // not by the segments seen: growing them meant hipFree + hipMalloc in the middle of the frame loop (a device-wide
        // synchronisation) while a pair held back by icelk_seg_track_defer could still point into the freed table.  A
        // segment with more tracks than rows fit simply takes no part.  Another row geometry (other LK parameters) is the
        // one case that allocates again: a waiting pair goes out first, and hipFree waits for whatever is in flight.
        if (!last && c->tmpl.row_bytes != per_row) {
            if (c->defer.pending) (void)flush_deferred(c);   // its job carries pointers into the tables about to go
            for (void*& b : c->tmpl.buf) {
                if (b) hipFree(b);
                b = nullptr;
            }
            c->tmpl.bytes = 0;
            c->tmpl.row_bytes = per_row;
            for (Ctx::SegBuf& o : c->sb) o.tmpl_for = -1;
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
            const size_t budget = std::min(c->tmpl.budget / 2, free_b / 4);   // per table
            const size_t rows = std::min((size_t)std::max(c->max_pts, 1), budget / per_row);
            const size_t alloc = rows * per_row;
            if (rows > 0 && hipMalloc(&c->tmpl.buf[0], alloc) == hipSuccess && hipMalloc(&c->tmpl.buf[1], alloc) == hipSuccess) {
                c->tmpl.bytes = alloc;
            } else {
                (void)hipGetLastError();
                if (c->tmpl.buf[0]) hipFree(c->tmpl.buf[0]);
                c->tmpl.buf[0] = c->tmpl.buf[1] = nullptr;
                c->tmpl.off = true;   // no room: every pair builds its own templates, as before (icelk_seg_template_info says so)
                c->tmpl.failed = true;
            }
        }
        if (c->tmpl.row_bytes != per_row) take = false;   // (the last pair of a segment with another row geometry)
        if (c->tmpl.bytes >= need && c->tmpl.row_bytes == per_row) {
            B.tmpl_levels = P.top_level + 1;
            if (take) {
                B.tmpl_in = c->tmpl.buf[set & 1];
                c->tmpl.taken++;
            }
            if (!last) {
                c->tmpl.left++;
                B.tmpl_out = c->tmpl.buf[set & 1];
                S.tmpl_for = S.vert + 1;
                S.tmpl_key = key;
                S.tmpl_slot = slot1;
                S.tmpl_gen = s1.gen;
            }
        }
    }
    return j;
}

// one event behind a tracker launch, for everything the launch read or wrote
static int record_launch(Ctx* c, hipEvent_t* ev)
{
    *ev = c->launch_ev[c->launch_seq++ % kLaunchEvents];
    HIPCHK(c, hipEventRecord(*ev, c->stream));
    return ICELK_OK;
}

static void seg_launched(Ctx* c, hipEvent_t ev, int set, int slot_prev, int slot_next)
{
    c->slots[slot_prev].used = ev;
    c->slots[slot_next].used = ev;
    c->sb[set].used = ev;
}

// a pair waiting for a partner goes out on its own
static int flush_deferred(Ctx* c)
{
    if (!c->defer.pending) return ICELK_OK;
    Ctx::Deferred& d = c->defer;
    d.pending = false;
    int rc;
    {
        ProfScope p(c, K_LK_FB);
        rc = launch_lk(c->stream, d.job.I, d.job.J, d.job.B, d.job.n, d.P, true);
    }
    if (rc) FAIL(c, rc, "unsupported window size");
    rc = check_launch(c, "lk_fb");
    if (rc) return rc;
    hipEvent_t ev;
    rc = record_launch(c, &ev);
    if (rc) return rc;
    seg_launched(c, ev, d.set, d.slot_prev, d.slot_next);
    return ICELK_OK;
}

// before a slot's frame or pyramid is overwritten: a waiting pair that reads it must have been launched
static int flush_deferred_slot(Ctx* c, int slot)
{
    if (c->defer.pending && (c->defer.slot_prev == slot || c->defer.slot_next == slot)) return flush_deferred(c);
    return ICELK_OK;
}

static bool same_lk_params(const LKParams& a, const LKParams& b)
{
    return a.win_w == b.win_w && a.win_h == b.win_h && a.top_level == b.top_level && a.max_count == b.max_count &&
           a.eps2 == b.eps2 && a.flags == b.flags && a.min_eig_thr == b.min_eig_thr && a.fb_thr == b.fb_thr &&
           a.margin == b.margin && a.dist_form == b.dist_form && a.sum_mode == b.sum_mode;
}

// shared by icelk_seg_track / icelk_seg_track_async / icelk_seg_track_defer
static int seg_track_core(Ctx* c, int slot_prev, int slot_next, int win_w, int win_h, int max_level, int crit_type,
                          int max_count, double epsilon, double min_eig_threshold, float fb_threshold, bool defer)
{
    Range rg(defer ? "icelk seg_track_defer" : "icelk seg_track (fused forward+backward LK launch)");
    int rc = check_slot(c, slot_prev, true);
    if (!rc) rc = check_slot(c, slot_next, true);
    if (rc) return rc;
    if (!c->seg_active) FAIL(c, ICELK_ESTATE, "icelk_seg_detect has not been called");
    Ctx::SegBuf& S = c->sb[c->sb_cur];
    if (S.vert >= kMaxVert) FAIL(c, ICELK_ECAP, "segment longer than the device track table");
    Slot& s0 = c->slots[slot_prev];
    Slot& s1 = c->slots[slot_next];
    if (s0.w != s1.w || s0.h != s1.h) FAIL(c, ICELK_EARG, "frame sizes differ");
    LKParams P;
    rc = make_lk_params(c, s0.w, s0.h, win_w, win_h, max_level, crit_type, max_count, epsilon, 0, min_eig_threshold,
                        fb_threshold, &P);
    if (rc) return rc;
    // a waiting pair of THIS segment comes first (pairs of a segment are sequential)
    if (c->defer.pending && c->defer.set == c->sb_cur) {
        rc = flush_deferred(c);
        if (rc) return rc;
    }
    rc = ensure_pyramid(c, slot_prev, P.top_level);
    if (!rc) rc = ensure_pyramid(c, slot_next, P.top_level);
    if (rc) return rc;
    rc = seg_wait(c);
    if (rc) return rc;
    if (S.upper > 0) {
        // tiles (half window + search margin) of a feature this close to the edge reach over it at the upper levels
        c->border_px = c->border_first ? ((std::max(win_w, win_h) / 2 + kLkTileMargin + 2) << std::max(P.top_level - 1, 0)) : 0;
        LKJob job = seg_job(c, c->sb_cur, s0, s1, P, true);
        // workgroup stamps describe ONE job: no pairing while they are on -- unless ICELK_LK_STAMPS_PAIR asks for the
        // stamps of a joint launch (indexed by workgroup: tools/lk_stamps_pair.py tells the jobs apart)
        static const bool stamp_pairs = getenv("ICELK_LK_STAMPS_PAIR") != nullptr;
        const bool diag = c->d_stamps != nullptr && !stamp_pairs;
        if (defer && !c->defer.pending && !diag) {
            // nothing goes out now: the pair waits for the first pair of the next segment (or another waiting pair)
            Ctx::Deferred& d = c->defer;
            d.pending = true;
            d.set = c->sb_cur;
            d.slot_prev = slot_prev;
            d.slot_next = slot_next;
            d.job = job;
            d.P = P;
            S.vert += 1;
            return ICELK_OK;
        }
        bool paired = false;
        if (c->defer.pending) {
            Ctx::Deferred& d = c->defer;
            if (!diag && same_lk_params(d.P, P)) {
                LKJob other = d.job;
                // the diagnostic arrays belong to the job of the current segment
                other.B.p_fwd = other.B.p_bwd = other.B.err_fwd = other.B.err_bwd = other.B.dist = nullptr;
                other.B.st_fwd = other.B.st_bwd = other.B.valid = nullptr;
                if (c->d_stamps && stamp_pairs && (size_t)(other.n + job.n + 32) <= c->stamps_cap) {
                    other.B.stamps = job.B.stamps = c->d_stamps;
                    hipMemsetAsync(c->d_stamps, 0, 3 * c->stamps_cap * 8, c->stream);
                }
                {
                    ProfScope p(c, K_LK_FB_PAIR);
                    paired = launch_lk_pair(c->stream, other, job, P);
                }
                if (paired) {
                    d.pending = false;
                    rc = check_launch(c, "lk_fb_pair");
                    if (rc) return rc;
                }
            }
            if (!paired) {
                rc = flush_deferred(c);
                if (rc) return rc;
            }
        }
        if (!paired) {
            LKBuffers& B = job.B;
            if (c->prof) {
                B.iters = c->d_iters;
                c->iters_n = S.upper;
                hipMemsetAsync(c->d_iters, 0xff, sizeof(uint32_t) * (size_t)S.upper, c->stream);   // dead tracks stay ~0
            }
            if (c->d_stamps && !stamp_pairs) {
                B.stamps = c->d_stamps;
                hipMemsetAsync(c->d_stamps, 0, 3 * c->stamps_cap * 8, c->stream);
            }
            {
                ProfScope p(c, K_LK_FB);
                rc = launch_lk(c->stream, job.I, job.J, B, job.n, P, true);
            }
            if (rc) FAIL(c, rc, "unsupported window size");
            rc = check_launch(c, "lk_fb");
            if (rc) return rc;
        }
        hipEvent_t ev;
        rc = record_launch(c, &ev);
        if (rc) return rc;
        if (paired) seg_launched(c, ev, c->defer.set, c->defer.slot_prev, c->defer.slot_next);
        seg_launched(c, ev, c->sb_cur, slot_prev, slot_next);
    } else if (c->defer.pending && !defer) {
        rc = flush_deferred(c);
        if (rc) return rc;
    }
    S.vert += 1;
    return ICELK_OK;
}

struct DevBufs {   // frees whatever was allocated when it goes out of scope
    std::vector<void*> p;
    ~DevBufs()
    {
        for (void* q : p)
            if (q) hipFree(q);
    }
    template <typename T>
    T* get(size_t count)
    {
        void* q = nullptr;
        if (hipMalloc(&q, sizeof(T) * (count ? count : 1)) != hipSuccess) return nullptr;
        p.push_back(q);
        return reinterpret_cast<T*>(q);
    }
};

}  // namespace icelk

using namespace icelk;

extern "C" {

int icelk_version(void) { return 100; }

const char* icelk_last_error(icelk_t* h)
{
    if (!h) return g_create_err.c_str();
    return C(h)->err.c_str();
}

int icelk_create(int device, int max_w, int max_h, int n_slots, int max_pts, icelk_t** out)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (!out || max_w <= 0 || max_h <= 0 || max_w > 65535 || max_h > 65535 || n_slots <= 0 || max_pts <= 0) {
        g_create_err = "icelk_create: bad argument";
        return ICELK_EARG;
    }
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
        g_create_err = std::string("icelk_create: no HIP device (") + hipGetErrorString(e) + ")";
        return ICELK_EHIP;
    }
    if (device < 0 || device >= ndev) {
        g_create_err = "icelk_create: device index out of range";
        return ICELK_EARG;
    }
    e = hipSetDevice(device);
    if (e != hipSuccess) {
        g_create_err = std::string("hipSetDevice: ") + hipGetErrorString(e);
        return ICELK_EHIP;
    }
    Ctx* c = new Ctx();
    c->device = device;
    c->max_w = max_w;
    c->max_h = max_h;
    c->n_slots = n_slots;
    c->max_pts = max_pts;
    int rc = ICELK_OK;
    auto fail = [&](int code) {
        g_create_err = c->err;
        destroy_ctx(c);
        return code;
    };
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess ||
        // uploads are DMA copies: normal priority; pyramid, detection and candidates are the high-priority streams
        hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->copy_stream2, hipStreamNonBlocking) != hipSuccess ||
        create_side_streams(c) != hipSuccess ||
        hipEventCreateWithFlags(&c->eo[0].done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->eo[1].done, hipEventDisableTiming) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&c->h_counts), 64, hipHostMallocMapped) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&c->h_seg), 64, hipHostMallocMapped) != hipSuccess ||
        hipEventCreateWithFlags(&c->det_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->corners_free, hipEventDisableTiming) != hipSuccess) {
        c->err = "hipStreamCreate failed";
        return fail(ICELK_EHIP);
    }
    memset(c->h_counts, 0, 64);
    c->stream = c->own_stream;
    c->slots.resize(n_slots);
    const size_t sb = slot_bytes(max_w, max_h);
    for (auto& s : c->slots) {
        if ((rc = dmalloc(c, &s.base, sb))) return fail(rc);
        s.bytes = sb;
        if (hipEventCreateWithFlags(&s.ready, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s.frame_ev, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s.used_own, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s.det_used, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s.eig_used, hipEventDisableTiming) != hipSuccess) {
            c->err = "hipEventCreate failed";
            return fail(ICELK_EHIP);
        }
        s.used = s.used_own;
        layout_levels(s, max_w, max_h);
        if (!layout_ok(s)) {
            c->err = "slot layout violates the dword-access invariant (internal)";
            return fail(ICELK_ECAP);
        }
    }
    for (auto& S : c->sb) {
        if (hipEventCreateWithFlags(&S.used_own, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&S.ready, hipEventDisableTiming) != hipSuccess) {
            c->err = "hipEventCreate failed";
            return fail(ICELK_EHIP);
        }
        S.used = S.used_own;
    }
    for (auto& e : c->launch_ev)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
            c->err = "hipEventCreate failed";
            return fail(ICELK_EHIP);
        }
    const size_t npx = (size_t)max_w * max_h;
    c->bgr_pitch = align_up(3 * max_w, kPitchAlign);
    c->mask_pitch = align_up(max_w, kPitchAlign);
    const size_t np = (size_t)max_pts;
    DetectScratch& D = c->D;
    D.cand_cap = (int)std::min<size_t>(candidate_capacity(max_w, max_h), (size_t)1 << 30);
    c->ncell_cap = npx + 1;
    D.sort_tmp_bytes = sort_tmp_bytes(D.cand_cap);
    if ((rc = dmalloc(c, &c->d_bgr, (size_t)c->bgr_pitch * max_h)) || (rc = dmalloc(c, &c->d_mask, (size_t)c->mask_pitch * max_h)) ||
        (rc = dmalloc(c, &c->d_p0, 2 * np)) || (rc = dmalloc(c, &c->d_p1, 2 * np)) || (rc = dmalloc(c, &c->d_p0r, 2 * np)) ||
        (rc = dmalloc(c, &c->d_err_f, np)) || (rc = dmalloc(c, &c->d_err_b, np)) || (rc = dmalloc(c, &c->d_dist, np)) ||
        (rc = dmalloc(c, &c->d_corners, 2 * np)) || (rc = dmalloc(c, &c->d_st_f, np)) || (rc = dmalloc(c, &c->d_st_b, np)) ||
        (rc = dmalloc(c, &c->d_valid, np)) || (rc = dmalloc(c, &D.eig, npx)) || (rc = dmalloc(c, &c->eo[0].max_key, 1)) || (rc = dmalloc(c, &c->eo[1].max_key, 1)) ||
        (rc = dmalloc(c, &c->eo[0].raw, (size_t)D.cand_cap)) || (rc = dmalloc(c, &c->eo[1].raw, (size_t)D.cand_cap)) ||
        (rc = dmalloc(c, &D.cand, (size_t)D.cand_cap)) || (rc = dmalloc(c, &D.cand_count, 1)) ||
        (rc = dmalloc(c, &D.cell_count, c->ncell_cap)) || (rc = dmalloc(c, &D.cell_start, c->ncell_cap)) ||
        (rc = dmalloc(c, &D.cell_fill, c->ncell_cap)) || (rc = dmalloc(c, &D.chunk_tot, (c->ncell_cap / 2048 + 2) * 32)) || (rc = dmalloc(c, &D.cell_cand, (size_t)D.cand_cap)) ||
        (rc = dmalloc(c, &D.state, (size_t)D.cand_cap)) || (rc = dmalloc(c, &D.undecided, 64)) ||
        (rc = dmalloc(c, &D.acc, (size_t)D.cand_cap)) ||
        (rc = dmalloc(c, &D.acc_sorted, (size_t)D.cand_cap)) ||
        (rc = dmalloc(c, &D.acc_count, 1)) || (rc = dmalloc(c, &c->eo[0].blk_count, candidate_blocks(max_w, max_h) * 4)) ||
        (rc = dmalloc(c, &c->eo[1].blk_count, candidate_blocks(max_w, max_h) * 4)) ||
        (rc = dmalloc(c, &D.key_hist, 1 << 16)) || (rc = dmalloc(c, &D.prune_key, 1)) || (rc = dmalloc(c, (uint8_t**)&D.sort_tmp, D.sort_tmp_bytes)) ||
        (rc = dmalloc(c, &D.tail_ctl, TC_WORDS_)) || (rc = dmalloc(c, &D.tail_resp, tail_resp_words())) ||
        (rc = dmalloc(c, &D.tail_bins, kTailOrderBins + 2)) ||
        (rc = dmalloc(c, &c->d_tracked, 64)) || (rc = dmalloc(c, &c->d_out_tracks, np * kMaxVert * 2)) ||
        (rc = dmalloc(c, &c->d_out_quality, np * (kMaxVert - 1))))
        return fail(rc);
    for (auto& S : c->sb)
        if ((rc = dmalloc(c, &S.live, 2 * np)) || (rc = dmalloc(c, &S.alive, np)) || (rc = dmalloc(c, &S.order, np)) ||
            (rc = dmalloc(c, &S.order_border, 1)) || (rc = dmalloc(c, &S.tracks, np * kMaxVert * 2)) ||
            (rc = dmalloc(c, &S.quality, np * (kMaxVert - 1))))
            return fail(rc);
    if (const char* ncs = getenv("ICELK_COPY_STREAMS")) {
        c->n_copy_streams = std::min(std::max(atoi(ncs), 1), 4);
        for (int k = 2; k < c->n_copy_streams; k++)
            if (hipStreamCreateWithFlags(&c->copy_more[k - 2], hipStreamNonBlocking) != hipSuccess) {
                c->err = "hipStreamCreate failed";
                return fail(ICELK_EHIP);
            }
    }
    c->use_order = getenv("ICELK_NO_ORDER") == nullptr;
    c->border_first = getenv("ICELK_NO_BORDER_FIRST") == nullptr;
    c->pyr_per_level = getenv("ICELK_PYR_PER_LEVEL") != nullptr;
    c->pyr_ahead_one_wave = getenv("ICELK_PYR_AHEAD_WIDE") == nullptr;
    c->tmpl.off = getenv("ICELK_NO_TEMPLATE_REUSE") != nullptr;
    if (const char* tb = getenv("ICELK_TEMPLATE_BUDGET_MB")) c->tmpl.budget = (size_t)std::max(atoll(tb), 0LL) << 20;
    c->host_tail = getenv("ICELK_HOST_TAIL") != nullptr;
    if (const char* fs = getenv("ICELK_TAIL_FORCE_STATUS")) c->tail_force_status = std::min(std::max(atoi(fs), 0), 4);
    if (!c->border_first) c->border_px = 0;
    if ((rc = dmalloc(c, &c->d_iters, (size_t)max_pts))) return fail(rc);
    if (const char* sp = getenv("ICELK_LK_STAMPS")) {
        c->stamps_path = sp;
        c->stamps_cap = (size_t)max_pts + 8;
        if ((rc = dmalloc(c, &c->d_stamps, 3 * c->stamps_cap))) return fail(rc);
    }
    if (const char* k = getenv("ICELK_LK_KERNEL")) {   // A/B measurements: "generic" | "multi" (default: one feature per wave)
        if (!strcmp(k, "generic")) c->lk_kernel_flags = ICELK_FLAG_GENERIC_KERNEL;
        else if (!strcmp(k, "multi")) c->lk_kernel_flags = ICELK_FLAG_MULTI_PER_WAVE;
    }
    // the third candidate buffer, and the second detector set (everything a detection in flight owns; the full-frame
    // eigenvalue map of icelk_min_eig_map is shared)
    if ((rc = dmalloc(c, &c->eo[2].max_key, 1)) || (rc = dmalloc(c, &c->eo[2].raw, (size_t)D.cand_cap)) ||
        (rc = dmalloc(c, &c->eo[2].blk_count, candidate_blocks(max_w, max_h) * 4)))
        return fail(rc);
    // the two-pass detector's scratch (three sets, ~20 MB each at 12 MP) only on a handle created with its switch set: a
    // handle without it runs the strip kernel whatever the switch says later (launch_candidates looks at D.acand)
    if (getenv("ICELK_TWO_PASS_CORNERS"))
      for (auto& e : c->eo)
        if ((rc = dmalloc(c, &e.acand, fast_cand_entries(max_w, max_h))) || (rc = dmalloc(c, &e.acount, fast_tiles(max_w, max_h))) ||
            (rc = dmalloc(c, &e.amaxc, fast_max_entries(max_w, max_h))) || (rc = dmalloc(c, &e.amaxn, fast_tiles(max_w, max_h))) ||
            (rc = dmalloc(c, &e.aemax, fast_tiles(max_w, max_h))) || (rc = dmalloc(c, &e.fmax_key, 8)) ||
            (rc = dmalloc(c, &e.aties, fast_cand_entries(max_w, max_h) + fast_tiles(max_w, max_h))))
            return fail(rc);
    if (hipEventCreateWithFlags(&c->eo[2].done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->counts_ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->tail_done, hipEventDisableTiming) != hipSuccess) {
        c->err = "hipEventCreate failed";
        return fail(ICELK_EHIP);
    }
    activate_eig_out(c, 0);
    det_save(c);                       // set 0 = what has been allocated so far
    {
        Ctx::DetSet& S = c->dset[1];
        DetectScratch& E = S.D;
        E = c->D;
        E.cand = nullptr; E.cand_count = nullptr; E.cell_count = nullptr; E.cell_start = nullptr; E.cell_fill = nullptr;
        E.chunk_tot = nullptr; E.cell_cand = nullptr; E.state = nullptr; E.undecided = nullptr; E.acc = nullptr;
        E.acc_sorted = nullptr; E.acc_count = nullptr; E.key_hist = nullptr; E.prune_key = nullptr; E.sort_tmp = nullptr;
        E.tail_ctl = nullptr; E.tail_resp = nullptr; E.tail_bins = nullptr;
        if ((rc = dmalloc(c, &E.cand, (size_t)D.cand_cap)) || (rc = dmalloc(c, &E.cand_count, 1)) ||
            (rc = dmalloc(c, &E.cell_count, c->ncell_cap)) || (rc = dmalloc(c, &E.cell_start, c->ncell_cap)) ||
            (rc = dmalloc(c, &E.cell_fill, c->ncell_cap)) || (rc = dmalloc(c, &E.chunk_tot, (c->ncell_cap / 2048 + 2) * 32)) ||
            (rc = dmalloc(c, &E.cell_cand, (size_t)D.cand_cap)) || (rc = dmalloc(c, &E.state, (size_t)D.cand_cap)) ||
            (rc = dmalloc(c, &E.undecided, 64)) || (rc = dmalloc(c, &E.acc, (size_t)D.cand_cap)) ||
            (rc = dmalloc(c, &E.acc_sorted, (size_t)D.cand_cap)) || (rc = dmalloc(c, &E.acc_count, 1)) ||
            (rc = dmalloc(c, &E.key_hist, 1 << 16)) || (rc = dmalloc(c, &E.prune_key, 1)) ||
            (rc = dmalloc(c, (uint8_t**)&E.sort_tmp, D.sort_tmp_bytes)) || (rc = dmalloc(c, &E.tail_ctl, TC_WORDS_)) ||
            (rc = dmalloc(c, &E.tail_resp, tail_resp_words())) || (rc = dmalloc(c, &E.tail_bins, kTailOrderBins + 2)))
            return fail(rc);
        if (hipMemset(E.tail_resp, 0, sizeof(int) * tail_resp_words()) != hipSuccess ||
            hipMemset(c->D.tail_resp, 0, sizeof(int) * tail_resp_words()) != hipSuccess ||
            hipMemset(E.tail_ctl, 0, sizeof(int) * TC_WORDS_) != hipSuccess ||
            hipMemset(E.tail_bins, 0, sizeof(int) * (kTailOrderBins + 2)) != hipSuccess ||
            hipMemset(c->D.tail_ctl, 0, sizeof(int) * TC_WORDS_) != hipSuccess ||
            hipMemset(c->D.tail_bins, 0, sizeof(int) * (kTailOrderBins + 2)) != hipSuccess) {
            c->err = "hipMemset failed";
            return fail(ICELK_EHIP);
        }
        if (hipHostMalloc(reinterpret_cast<void**>(&S.h_counts), 64, hipHostMallocMapped) != hipSuccess ||
            hipEventCreateWithFlags(&S.counts_ev, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&S.tail_done, hipEventDisableTiming) != hipSuccess) {
            c->err = "hipHostMalloc failed";
            return fail(ICELK_EHIP);
        }
        memset(S.h_counts, 0, 64);
        S.eo_active = 1;
        E.raw = c->eo[1].raw;
        E.blk_count = c->eo[1].blk_count;
        E.max_key = c->eo[1].max_key;
        E.acand = c->eo[1].acand; E.acount = c->eo[1].acount; E.amaxc = c->eo[1].amaxc; E.amaxn = c->eo[1].amaxn;
        E.aemax = c->eo[1].aemax; E.fmax_key = c->eo[1].fmax_key; E.aties = c->eo[1].aties;
    }
    if (hipMemset(c->d_tracked, 0, 64 * 8) != hipSuccess) {
        c->err = "hipMemset failed";
        return fail(ICELK_EHIP);
    }
    // The host's tail (detect_finish) sorts with rocPRIM, and the first launch of its kernels in a process costs the host
    // ~8 ms (the code object of k_sort.hip is loaded then).  With the device-driven tail that first time was some
    // detection in the MIDDLE of a run -- the first one the device handed back -- and a 64-pair batch lasted 26 ms instead
    // of 17 (profiles/r04_c3_stall.txt).  Paid here instead: 64 keys through the sort, once per handle.
    if (hipMemsetAsync(c->D.acc, 0, 64 * sizeof(unsigned long long), c->tail_stream) != hipSuccess) {
        c->err = "hipMemset failed";
        return fail(ICELK_EHIP);
    }
    sort_keys_desc(c->tail_stream, c->D, c->D.acc, c->D.acc_sorted, 64);
    if (hipStreamSynchronize(c->tail_stream) != hipSuccess) {
        c->err = "warm-up sort failed";
        return fail(ICELK_EHIP);
    }
    *out = reinterpret_cast<icelk_t*>(c);
    return ICELK_OK;
}

int icelk_destroy(icelk_t* h)
{
    if (!h) return ICELK_EARG;
    destroy_ctx(C(h));
    return ICELK_OK;
}

int icelk_set_stream(icelk_t* h, void* hip_stream)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : c->own_stream;
    return ICELK_OK;
}

int icelk_sync(icelk_t* h)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    int rcf = flush_deferred(c);   // a pair waiting for a partner counts as issued work
    if (rcf) return rcf;
    // every stream of the handle: uploads / pyramids built ahead, candidate kernels of a prepared detection
    // (icelk_seg_detect_prepare), the min-distance / sort / emit stage, tracker launches
    HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    HIPCHK(c, hipStreamSynchronize(c->copy_stream2));
    for (auto q : c->copy_hi)
        if (q) HIPCHK(c, hipStreamSynchronize(q));
    for (auto q : c->copy_more)
        if (q) HIPCHK(c, hipStreamSynchronize(q));
    HIPCHK(c, hipStreamSynchronize(c->pyr_stream));
    HIPCHK(c, hipStreamSynchronize(c->eig_stream));
    HIPCHK(c, hipStreamSynchronize(c->det_stream));
    HIPCHK(c, hipStreamSynchronize(c->tail_stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ICELK_OK;
}

int icelk_set_lk_kernel(icelk_t* h, int which)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    if (which != 0 && which != ICELK_FLAG_GENERIC_KERNEL && which != ICELK_FLAG_MULTI_PER_WAVE)
        FAIL(c, ICELK_EARG, "bad kernel selector");
    c->lk_kernel_flags = which;
    return ICELK_OK;
}

int icelk_set_variant(icelk_t* h, const char* name, int value)
{
    if (!h || !name) return ICELK_EARG;
    Ctx* c = C(h);
    if (!strcmp(name, "lk_sums") && value >= 0 && value <= 2) c->lk_sum_mode = value;
    else if (!strcmp(name, "sobel_fma") && value >= 0 && value <= 3) c->corner_variant = (c->corner_variant & 4) | value;
    else if (!strcmp(name, "eig_fma") && (value == 0 || value == 1)) c->corner_variant = (c->corner_variant & 3) | (value << 2);
    else FAIL(c, ICELK_EARG, "unknown variant / value");
    for (auto& e : c->eo) e.valid = false;     // candidates prepared under another variant are not adopted
    return ICELK_OK;
}

int icelk_set_fb_distance(icelk_t* h, int form)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    if (form != ICELK_FB_HYPOT && form != ICELK_FB_SQRT) FAIL(c, ICELK_EARG, "bad forward-backward distance form");
    c->fb_dist_form = form;
    return ICELK_OK;
}

// ---- ingest ------------------------------------------------------------------------------------
int icelk_upload_gray(icelk_t* h, int slot, const uint8_t* host, int w, int h_, int stride)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    if (!host || stride < w) FAIL(c, ICELK_EARG, "bad host image");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = begin_frame(c, slot, w, h_);
    if (rc) return rc;
    Slot& s = c->slots[slot];
    HIPCHK(c, hipMemcpy2DAsync(s.lv[0].ptr, s.lv[0].pitch, host, stride, w, h_, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipEventRecord(s.frame_ev, c->stream));
    s.levels_built = 1;
    s.pending = false;
    return ICELK_OK;
}

int icelk_upload_gray_async(icelk_t* h, int slot, const uint8_t* pinned_host, int w, int h_, int stride)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    Range rg("icelk upload_gray_async");
    if (!pinned_host || stride < w) FAIL(c, ICELK_EARG, "bad host image");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = begin_frame(c, slot, w, h_);
    if (rc) return rc;
    Slot& s = c->slots[slot];
    // Two streams in turn (see Ctx::copy_stream), of the high-priority class.  With streams of the compute stream's own class
    // every second upload -- always those of ONE of the two streams -- started 60-180 us after the copy before it had
    // ended (profiles/r04_c3_modes.txt): that stream shared its hardware queue with the compute stream, and the barrier
    // that carries an upload's dependencies stood behind a 250-us tracker launch.  C3 with 6 uploads in flight:
    // 3 650-3 800 -> 4 040-4 110 pairs/s (profiles/r04_c3_copy_prio.txt).  ICELK_COPY_PRIORITY=normal: the streams of before.
    static const bool copy_normal = getenv("ICELK_COPY_PRIORITY") && !strcmp(getenv("ICELK_COPY_PRIORITY"), "normal");
    const unsigned useq = c->upload_seq++ % (unsigned)c->n_copy_streams;
    if (!copy_normal && useq < 2 && !c->copy_hi[useq])
        HIPCHK(c, create_priority_stream(&c->copy_hi[useq]));
    const hipStream_t cs = useq >= 2 ? c->copy_more[useq - 2] : (!copy_normal ? c->copy_hi[useq] : (useq == 0 ? c->copy_stream : c->copy_stream2));
    // the copy must not overtake the launches that still read this slot (Slot::used / det_used)
    if (int rcw = wait_event(c, cs, s.used)) return rcw;
    if (s.pending) if (int rcw = wait_event(c, cs, s.ready)) return rcw;   // an upload or a pyramid built ahead still in flight
    if (int rcw = wait_event(c, cs, s.det_used)) return rcw;
    if (int rcw = wait_event(c, cs, s.eig_used)) return rcw;
    HIPCHK(c, hipMemcpy2DAsync(s.lv[0].ptr, s.lv[0].pitch, pinned_host, stride, w, h_, hipMemcpyHostToDevice, cs));
    HIPCHK(c, hipEventRecord(s.ready, cs));
    HIPCHK(c, hipEventRecord(s.frame_ev, cs));
    s.pending = true;
    s.levels_built = 1;
    return ICELK_OK;
}

int icelk_host_alloc(void** out, uint64_t bytes)
{
    if (!out) return ICELK_EARG;
    return hipHostMalloc(out, bytes, hipHostMallocDefault) == hipSuccess ? ICELK_OK : ICELK_ENOMEM;
}

int icelk_host_free(void* p) { return hipHostFree(p) == hipSuccess ? ICELK_OK : ICELK_EHIP; }

int icelk_upload_bgr(icelk_t* h, int slot, const uint8_t* host, int w, int h_, int stride, int gray_variant)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    if (!host || stride < 3 * w) FAIL(c, ICELK_EARG, "bad host image");
    if (gray_variant != ICELK_GRAY_CV3 && gray_variant != ICELK_GRAY_CV4) FAIL(c, ICELK_EARG, "bad gray variant");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = begin_frame(c, slot, w, h_);
    if (rc) return rc;
    Slot& s = c->slots[slot];
    HIPCHK(c, hipMemcpy2DAsync(c->d_bgr, c->bgr_pitch, host, stride, 3 * (size_t)w, h_, hipMemcpyHostToDevice, c->stream));
    {
        ProfScope p(c, K_GRAY);
        launch_bgr2gray(c->stream, c->d_bgr, c->bgr_pitch, s.lv[0].ptr, s.lv[0].pitch, w, h_, gray_variant);
    }
    rc = check_launch(c, "bgr2gray");
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipEventRecord(s.frame_ev, c->stream));
    s.levels_built = 1;
    s.pending = false;
    return ICELK_OK;
}

int icelk_set_gray_device(icelk_t* h, int slot, const void* dev, int w, int h_, int stride)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    if (!dev || stride < w) FAIL(c, ICELK_EARG, "bad device image");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = begin_frame(c, slot, w, h_);
    if (rc) return rc;
    Slot& s = c->slots[slot];
    HIPCHK(c, hipMemcpy2DAsync(s.lv[0].ptr, s.lv[0].pitch, dev, stride, w, h_, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipEventRecord(s.frame_ev, c->stream));
    s.levels_built = 1;
    s.pending = false;
    return ICELK_OK;
}

int icelk_cvt_bgr_device(icelk_t* h, int slot, const void* dev_bgr, int w, int h_, int stride, int gray_variant)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    if (!dev_bgr || stride < 3 * w) FAIL(c, ICELK_EARG, "bad device image");
    if (gray_variant != ICELK_GRAY_CV3 && gray_variant != ICELK_GRAY_CV4) FAIL(c, ICELK_EARG, "bad gray variant");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = begin_frame(c, slot, w, h_);
    if (rc) return rc;
    Slot& s = c->slots[slot];
    {
        ProfScope p(c, K_GRAY);
        launch_bgr2gray(c->stream, reinterpret_cast<const uint8_t*>(dev_bgr), stride, s.lv[0].ptr, s.lv[0].pitch, w, h_,
                        gray_variant);
    }
    rc = check_launch(c, "bgr2gray");
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(s.frame_ev, c->stream));
    s.levels_built = 1;
    s.pending = false;
    return ICELK_OK;
}

int icelk_synth_frame_affine(icelk_t* h, int slot, int w, int h_, int64_t ux, int64_t uy, uint32_t seed,
                             const int32_t* affine)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    if (affine)
        for (int k = 0; k < 4; k++)
            if (affine[k] > (1 << 13) || affine[k] < -(1 << 13)) FAIL(c, ICELK_EARG, "affine coefficient beyond +-2^-7");
    int rc = begin_frame(c, slot, w, h_);
    if (rc) return rc;
    Slot& s = c->slots[slot];
    {
        ProfScope p(c, K_SYNTH);
        launch_synth(c->stream, s.lv[0], ux, uy, seed, affine);
    }
    rc = check_launch(c, "synth");
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(s.frame_ev, c->stream));
    s.levels_built = 1;
    s.pending = false;
    return ICELK_OK;
}

int icelk_synth_frame(icelk_t* h, int slot, int w, int h_, int64_t ux, int64_t uy, uint32_t seed)
{
    return icelk_synth_frame_affine(h, slot, w, h_, ux, uy, seed, nullptr);
}

int icelk_drop_pyramid(icelk_t* h, int slot)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    int rc = check_slot(c, slot, true);
    if (!rc) rc = flush_deferred_slot(c, slot);
    if (rc) return rc;
    c->slots[slot].levels_built = 1;
    return ICELK_OK;
}

int icelk_download_level(icelk_t* h, int slot, int level, uint8_t* host, int stride, int* w, int* h_)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    int rc = check_slot(c, slot, true);
    if (rc) return rc;
    Slot& s = c->slots[slot];
    if (level < 0 || level >= s.levels_built) FAIL(c, ICELK_ESTATE, "pyramid level not built");
    rc = wait_slot(c, slot);
    if (rc) return rc;
    const Level& L = s.lv[level];
    if (w) *w = L.w;
    if (h_) *h_ = L.h;
    if (host) {
        if (stride < L.w) FAIL(c, ICELK_EARG, "stride smaller than the level width");
        HIPCHK(c, hipMemcpy2DAsync(host, stride, L.ptr, L.pitch, L.w, L.h, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return ICELK_OK;
}

int icelk_build_pyramid(icelk_t* h, int slot, int win_w, int win_h, int max_level, int* out_levels)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    int rc = check_slot(c, slot, true);
    if (rc) return rc;
    if (win_w <= 2 || win_h <= 2 || max_level < 0) FAIL(c, ICELK_EARG, "bad pyramid parameters");
    if (max_level > kMaxLevels - 1) max_level = kMaxLevels - 1;
    Slot& s = c->slots[slot];
    const int top = pyramid_top_level(s.w, s.h, win_w, win_h, max_level);
    rc = ensure_pyramid(c, slot, top);
    if (rc) return rc;
    if (out_levels) *out_levels = top;
    return ICELK_OK;
}

int icelk_build_pyramid_ahead(icelk_t* h, int slot, int win_w, int win_h, int max_level)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    Range rg("icelk build_pyramid_ahead");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = check_slot(c, slot, true);
    if (rc) return rc;
    if (win_w <= 2 || win_h <= 2 || max_level < 0) FAIL(c, ICELK_EARG, "bad pyramid parameters");
    if (max_level > kMaxLevels - 1) max_level = kMaxLevels - 1;
    Slot& s = c->slots[slot];
    const int top = pyramid_top_level(s.w, s.h, win_w, win_h, max_level);
    if (s.levels_built >= top + 1) return ICELK_OK;
    const hipStream_t cs = c->pyr_stream;
    // level 0 must be there (it may have been written on the compute stream), and launches that still read the
    // slot's previous pyramid must be through
    if (int rcw = wait_event(c, cs, s.frame_ev)) return rcw;
    if (s.pending) if (int rcw = wait_event(c, cs, s.ready)) return rcw;
    if (int rcw = wait_event(c, cs, s.used)) return rcw;
    rc = build_levels(c, s, top, cs);
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(s.ready, cs));
    s.pending = true;
    return ICELK_OK;
}

// ---- tracker -----------------------------------------------------------------------------------
int icelk_pyrlk(icelk_t* h, int prev_slot, int next_slot, const float* prev_xy, float* next_xy, uint8_t* status,
                float* err, int n, int win_w, int win_h, int max_level, int crit_type, int max_count, double epsilon,
                int flags, double min_eig_threshold)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    int rc = check_slot(c, prev_slot, true);
    if (!rc) rc = check_slot(c, next_slot, true);
    if (rc) return rc;
    if (n < 0) FAIL(c, ICELK_EARG, "negative point count");
    if (n > c->max_pts) FAIL(c, ICELK_ECAP, "more points than max_pts of icelk_create");
    Slot& s0 = c->slots[prev_slot];
    Slot& s1 = c->slots[next_slot];
    if (s0.w != s1.w || s0.h != s1.h) FAIL(c, ICELK_EARG, "frame sizes differ");
    LKParams P;
    rc = make_lk_params(c, s0.w, s0.h, win_w, win_h, max_level, crit_type, max_count, epsilon, flags, min_eig_threshold,
                        1.f, &P);
    if (rc) return rc;
    if (n == 0) return ICELK_OK;
    if (!prev_xy || !next_xy) FAIL(c, ICELK_EARG, "null point buffer");
    rc = ensure_pyramid(c, prev_slot, P.top_level);
    if (!rc) rc = ensure_pyramid(c, next_slot, P.top_level);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_p0, prev_xy, sizeof(float) * 2 * n, hipMemcpyHostToDevice, c->stream));
    if (flags & ICELK_FLAG_INITIAL_FLOW)
        HIPCHK(c, hipMemcpyAsync(c->d_p1, next_xy, sizeof(float) * 2 * n, hipMemcpyHostToDevice, c->stream));
    LKBuffers B{};
    B.p_in = c->d_p0;
    B.p_fwd = c->d_p1;
    B.st_fwd = c->d_st_f;
    B.err_fwd = err ? c->d_err_f : nullptr;
    if (c->prof) { B.iters = c->d_iters; c->iters_n = n; }
    {
        ProfScope p(c, K_LK);
        rc = launch_lk(c->stream, pyramid_of(s0), pyramid_of(s1), B, n, P, false);
    }
    if (rc) FAIL(c, rc, "unsupported window size");
    rc = check_launch(c, "lk");
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(next_xy, c->d_p1, sizeof(float) * 2 * n, hipMemcpyDeviceToHost, c->stream));
    if (status) HIPCHK(c, hipMemcpyAsync(status, c->d_st_f, n, hipMemcpyDeviceToHost, c->stream));
    if (err) HIPCHK(c, hipMemcpyAsync(err, c->d_err_f, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ICELK_OK;
}

int icelk_track_fb(icelk_t* h, int slot0, int slot1, const float* p0, int n, int win_w, int win_h, int max_level,
                   int crit_type, int max_count, double epsilon, double min_eig_threshold, float fb_threshold,
                   float* p1, float* p0r, uint8_t* st_fwd, uint8_t* st_bwd, float* err_fwd, float* err_bwd, float* dist,
                   uint8_t* valid)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    int rc = check_slot(c, slot0, true);
    if (!rc) rc = check_slot(c, slot1, true);
    if (rc) return rc;
    if (n < 0) FAIL(c, ICELK_EARG, "negative point count");
    if (n > c->max_pts) FAIL(c, ICELK_ECAP, "more points than max_pts of icelk_create");
    Slot& s0 = c->slots[slot0];
    Slot& s1 = c->slots[slot1];
    if (s0.w != s1.w || s0.h != s1.h) FAIL(c, ICELK_EARG, "frame sizes differ");
    LKParams P;
    rc = make_lk_params(c, s0.w, s0.h, win_w, win_h, max_level, crit_type, max_count, epsilon, 0, min_eig_threshold,
                        fb_threshold, &P);
    if (rc) return rc;
    if (n == 0) return ICELK_OK;
    if (!p0) FAIL(c, ICELK_EARG, "null point buffer");
    rc = ensure_pyramid(c, slot0, P.top_level);
    if (!rc) rc = ensure_pyramid(c, slot1, P.top_level);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_p0, p0, sizeof(float) * 2 * n, hipMemcpyHostToDevice, c->stream));
    LKBuffers B{};
    B.p_in = c->d_p0;
    B.p_fwd = c->d_p1;
    B.st_fwd = c->d_st_f;
    B.err_fwd = err_fwd ? c->d_err_f : nullptr;   // the residual error is only formed for a caller that takes it
    B.p_bwd = c->d_p0r;
    B.st_bwd = c->d_st_b;
    B.err_bwd = err_bwd ? c->d_err_b : nullptr;
    B.dist = c->d_dist;
    B.valid = c->d_valid;
    if (c->prof) { B.iters = c->d_iters; c->iters_n = n; }
    {
        ProfScope p(c, K_LK_FB);
        rc = launch_lk(c->stream, pyramid_of(s0), pyramid_of(s1), B, n, P, true);
    }
    if (rc) FAIL(c, rc, "unsupported window size");
    rc = check_launch(c, "lk_fb");
    if (rc) return rc;
    const size_t fb = sizeof(float) * n;
    if (p1) HIPCHK(c, hipMemcpyAsync(p1, c->d_p1, 2 * fb, hipMemcpyDeviceToHost, c->stream));
    if (p0r) HIPCHK(c, hipMemcpyAsync(p0r, c->d_p0r, 2 * fb, hipMemcpyDeviceToHost, c->stream));
    if (st_fwd) HIPCHK(c, hipMemcpyAsync(st_fwd, c->d_st_f, n, hipMemcpyDeviceToHost, c->stream));
    if (st_bwd) HIPCHK(c, hipMemcpyAsync(st_bwd, c->d_st_b, n, hipMemcpyDeviceToHost, c->stream));
    if (err_fwd) HIPCHK(c, hipMemcpyAsync(err_fwd, c->d_err_f, fb, hipMemcpyDeviceToHost, c->stream));
    if (err_bwd) HIPCHK(c, hipMemcpyAsync(err_bwd, c->d_err_b, fb, hipMemcpyDeviceToHost, c->stream));
    if (dist) HIPCHK(c, hipMemcpyAsync(dist, c->d_dist, fb, hipMemcpyDeviceToHost, c->stream));
    if (valid) HIPCHK(c, hipMemcpyAsync(valid, c->d_valid, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ICELK_OK;
}

int icelk_fb_filter(icelk_t* h, const float* p0, const float* p0r, int n, float fb_threshold, float* dist,
                    uint8_t* valid)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    if (n < 0) FAIL(c, ICELK_EARG, "negative point count");
    if (n > c->max_pts) FAIL(c, ICELK_ECAP, "more points than max_pts of icelk_create");
    if (n == 0) return ICELK_OK;
    if (!p0 || !p0r) FAIL(c, ICELK_EARG, "null point buffer");
    HIPCHK(c, hipMemcpyAsync(c->d_p0, p0, sizeof(float) * 2 * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_p0r, p0r, sizeof(float) * 2 * n, hipMemcpyHostToDevice, c->stream));
    launch_fb_filter(c->stream, c->d_p0, c->d_p0r, n, fb_threshold, c->fb_dist_form, c->d_dist, c->d_valid);
    int rc = check_launch(c, "fb_filter");
    if (rc) return rc;
    if (dist) HIPCHK(c, hipMemcpyAsync(dist, c->d_dist, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
    if (valid) HIPCHK(c, hipMemcpyAsync(valid, c->d_valid, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ICELK_OK;
}

// ---- detector ----------------------------------------------------------------------------------
int icelk_set_mask(icelk_t* h, const uint8_t* host_mask, int w, int h_, int stride)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    if (!host_mask) {
        c->has_mask = false;
        c->mask_gen++;
        return ICELK_OK;
    }
    if (w <= 0 || h_ <= 0 || stride < w) FAIL(c, ICELK_EARG, "bad mask");
    if (w > c->max_w || h_ > c->max_h) FAIL(c, ICELK_ECAP, "mask larger than max_w x max_h");
    HIPCHK(c, hipStreamSynchronize(c->det_stream));
    HIPCHK(c, hipStreamSynchronize(c->eig_stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy2DAsync(c->d_mask, c->mask_pitch, host_mask, stride, w, h_, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->has_mask = true;
    c->mask_gen++;
    c->mask_w = w;
    c->mask_h = h_;
    return ICELK_OK;
}

int icelk_set_mask_polygon(icelk_t* h, const double* poly_xy, int n, double crop_left, double crop_top, int w, int h_)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    if (n < 0 || n > 65536 || (n > 0 && !poly_xy) || w <= 0 || h_ <= 0) FAIL(c, ICELK_EARG, "bad polygon / frame size");
    if (w > c->max_w || h_ > c->max_h) FAIL(c, ICELK_ECAP, "mask larger than max_w x max_h");
    // no detection may be reading the old mask
    HIPCHK(c, hipStreamSynchronize(c->det_stream));
    HIPCHK(c, hipStreamSynchronize(c->eig_stream));
    double* d_poly = nullptr;
    int rc = dmalloc(c, &d_poly, 2 * (size_t)(n > 0 ? n : 1));
    if (rc) return rc;
    hipError_t e = n > 0 ? hipMemcpyAsync(d_poly, poly_xy, sizeof(double) * 2 * n, hipMemcpyHostToDevice, c->stream)
                         : hipSuccess;
    if (e == hipSuccess) {
        launch_polygon_mask(c->stream, d_poly, n, crop_left, crop_top, w, h_, c->d_mask, c->mask_pitch);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d_poly);
    if (e != hipSuccess) {
        c->err = std::string("polygon mask: ") + hipGetErrorString(e);
        return ICELK_EHIP;
    }
    c->has_mask = true;
    c->mask_gen++;
    c->mask_w = w;
    c->mask_h = h_;
    return ICELK_OK;
}

int icelk_download_mask(icelk_t* h, uint8_t* host_mask, int stride, int* w, int* h_)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->has_mask) FAIL(c, ICELK_ESTATE, "no mask set");
    if (w) *w = c->mask_w;
    if (h_) *h_ = c->mask_h;
    if (!host_mask) return ICELK_OK;
    if (stride < c->mask_w) FAIL(c, ICELK_EARG, "bad host stride");
    HIPCHK(c, hipMemcpy2DAsync(host_mask, stride, c->d_mask, c->mask_pitch, c->mask_w, c->mask_h, hipMemcpyDeviceToHost,
                               c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ICELK_OK;
}

int icelk_min_eig_map(icelk_t* h, int slot, int block_size, float* host_out, int stride_elems)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    int rc = check_slot(c, slot, true);
    if (rc) return rc;
    Slot& s = c->slots[slot];
    if (!host_out || stride_elems < s.w || block_size <= 0) FAIL(c, ICELK_EARG, "bad argument");
    if (min_eig_lds_bytes(block_size) > 150 * 1024) FAIL(c, ICELK_EARG, "blockSize too large");
    rc = wait_slot(c, slot);
    if (rc) return rc;
    {
        const int k = det_free(c);     // scratch of a set no detection in flight owns
        if (k < 0) FAIL(c, ICELK_ESTATE, "two detections are in flight");
        det_load(c, k);
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));       // the frame is in place
    HIPCHK(c, hipStreamSynchronize(c->det_stream));   // the detector scratch is free
    HIPCHK(c, hipStreamSynchronize(c->tail_stream));
    {
        ProfScope p(c, K_EIG);
        launch_detect_reset(c->stream, c->D, 0, 3);
        c->counters_clean = false;
        if (fused_block_size(block_size) && !getenv("ICELK_GENERIC_CORNERS") && !c->corner_variant) {
            launch_candidates(c->stream, c->D, s.lv[0], block_size, nullptr, 0, 1.0, false, c->D.eig);
        } else {
            launch_min_eig(c->stream, s.lv[0], block_size, c->D.eig, nullptr, 0, c->D.max_key, c->corner_variant);
        }
    }
    rc = check_launch(c, "min_eig");
    if (rc) return rc;
    HIPCHK(c, hipMemcpy2DAsync(host_out, sizeof(float) * stride_elems, c->D.eig, sizeof(float) * s.w, sizeof(float) * s.w,
                               s.h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ICELK_OK;
}

int icelk_good_features(icelk_t* h, int slot, int use_mask, int max_corners, double quality_level, double min_distance,
                        int block_size, float* out_xy, int cap, int* out_n)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    if (!out_n || cap < 0 || (cap > 0 && !out_xy)) FAIL(c, ICELK_EARG, "bad output buffer");
    int n = 0;
    int rc = detect_core(c, slot, use_mask, max_corners, quality_level, min_distance, block_size, cap, &n);
    if (rc) return rc;
    if (n > 0) {
        HIPCHK(c, hipMemcpyAsync(out_xy, c->d_corners, sizeof(float) * 2 * n, hipMemcpyDeviceToHost, c->tail_stream));
        HIPCHK(c, hipStreamSynchronize(c->tail_stream));
    }
    *out_n = n;
    return ICELK_OK;
}

int icelk_detect_stats(icelk_t* h, int* n_candidates, int* n_accepted)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    if (n_candidates) *n_candidates = c->last_candidates;
    if (n_accepted) *n_accepted = c->last_accepted;
    return ICELK_OK;
}

int icelk_detect_fast_stats(icelk_t* h, int slot_w, int slot_h, long long* out)
{
    if (!h || !out) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    det_save(c);
    const Ctx::EigOut& e = c->eo[c->eo_active];
    if (!e.acand) FAIL(c, ICELK_EARG, "the two-pass detector was not enabled when this handle was created (ICELK_TWO_PASS_CORNERS)");
    const size_t nt = fast_tiles(slot_w, slot_h);
    std::vector<int> cnt(nt), mx(nt);
    unsigned fk[3] = {0, 0, 0};
    HIPCHK(c, hipMemcpy(cnt.data(), e.acount, nt * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(mx.data(), e.amaxn, nt * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(fk, e.fmax_key, sizeof fk, hipMemcpyDeviceToHost));
    long long listed = 0, whole = 0, maxc = 0, maxover = 0, biggest = 0;
    for (size_t i = 0; i < nt; i++) {
        if (cnt[i] > 256) whole++;
        else listed += cnt[i];
        if (cnt[i] > biggest) biggest = cnt[i];
        if (mx[i] > 16) maxover++;
        else maxc += mx[i];
    }
    out[0] = (long long)nt; out[1] = listed; out[2] = whole; out[3] = (long long)fk[1]; out[4] = maxc; out[5] = maxover;
    out[6] = biggest; out[7] = (long long)fk[2];
    return ICELK_OK;
}

// ---- segment state -----------------------------------------------------------------------------
int icelk_seg_detect_begin(icelk_t* h, int slot, int use_mask, int max_corners, double quality_level,
                            double min_distance, int block_size)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    return detect_begin(c, slot, use_mask, max_corners, quality_level, min_distance, block_size, true);
}

int icelk_seg_detect_prepare(icelk_t* h, int slot, int use_mask, int block_size)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    if (block_size <= 0) FAIL(c, ICELK_EARG, "bad detector parameters");
    return detect_prepare(c, slot, use_mask, block_size);
}

// The corners of the detection in flight become a new segment in the OTHER set of segment buffers (the current one
// keeps being tracked); icelk_seg_switch makes it current.
static int seg_stage(Ctx* c, int max_corners, int* out_n)
{
    if (c->seg_staged) FAIL(c, ICELK_ESTATE, "a staged segment is waiting for icelk_seg_switch");
    int n = 0;
    // the new segment goes into the set after the current one, on the tail stream right behind the corner list
    Ctx::SegBuf& nb = c->sb[(c->sb_cur + 1) % kSegSets];
    bool tables_written = false, dev_done = false;
    int rc = detect_finish(c, max_corners, c->max_pts, &n, &nb, &tables_written, &dev_done);
    if (rc) return rc;
    if (!dev_done) {
        const hipStream_t ds = c->tail_stream;
        if (!tables_written) {
            // launches that still touch that set (a segment closed several switches ago) must be through
            if (int rcw = wait_event(c, ds, nb.used)) return rcw;
            launch_seg_init(ds, c->d_corners, n, nb.live, nb.alive, nb.tracks, kMaxVert);
        }
        if (c->use_order) launch_seg_order(ds, c->d_corners, n, c->job.w, c->job.h, c->border_px, nb.order, nb.order_border);
        rc = check_launch(c, "seg_init");
        if (rc) return rc;
        HIPCHK(c, hipEventRecord(c->corners_free, ds));
        HIPCHK(c, hipEventRecord(nb.ready, ds));
    }
    c->seg_staged = true;
    c->staged_n = n;
    if (out_n) *out_n = n;
    return ICELK_OK;
}

static int seg_switch(Ctx* c)
{
    if (!c->seg_staged) FAIL(c, ICELK_ESTATE, "no staged segment (icelk_seg_detect_stage has not been called)");
    // a pair still waiting from before the previous switch has found no partner
    if (c->defer.pending && c->defer.set != c->sb_cur) {
        int rc = flush_deferred(c);
        if (rc) return rc;
    }
    c->seg_staged = false;
    c->closed_valid = c->seg_active;
    c->sb_cur = (c->sb_cur + 1) % kSegSets;
    c->seg_ready_pending = true;
    c->sb[c->sb_cur].vert = 1;
    c->sb[c->sb_cur].tmpl_for = -1;
    c->sb[c->sb_cur].upper = c->staged_n;
    c->seg_active = true;
    return ICELK_OK;
}

int icelk_seg_detect_stage(icelk_t* h, int max_corners, int* out_n)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    return seg_stage(c, max_corners, out_n);
}

int icelk_seg_detect_stage_try(icelk_t* h, int max_corners, int* out_n, int* out_done)
{
    if (!h || !out_done) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    *out_done = 0;
    if (c->seg_staged) FAIL(c, ICELK_ESTATE, "a staged segment is waiting for icelk_seg_switch");
    const int k = det_oldest(c);
    if (k < 0) FAIL(c, ICELK_ESTATE, "no detection in flight");
    // the counts of the oldest detection in flight: published behind its min-distance stage (publish_counts)
    if (!counts_here(c->dset[k].h_counts, c->dset[k].counts_seq)) {
        const hipError_t q = hipEventQuery(c->dset[k].counts_ev);
        if (q == hipErrorNotReady) {
            (void)hipGetLastError();
            return ICELK_OK;
        }
        HIPCHK(c, q);
    }
    int rc = seg_stage(c, max_corners, out_n);
    if (rc) return rc;
    *out_done = 1;
    return ICELK_OK;
}

int icelk_seg_switch(icelk_t* h)
{
    if (!h) return ICELK_EARG;
    return seg_switch(C(h));
}

int icelk_seg_detect_cancel(icelk_t* h)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    // whatever was enqueued for the abandoned detections runs to its end: nothing is left reading a slot or a mask
    HIPCHK(c, hipStreamSynchronize(c->eig_stream));
    HIPCHK(c, hipStreamSynchronize(c->det_stream));
    HIPCHK(c, hipStreamSynchronize(c->tail_stream));
    det_save(c);
    for (auto& S : c->dset)
        if (S.job.active) {
            S.job.active = false;
            S.counters_clean = false;   // its counters were never reset by a tail: the next detection of the set resets them
        }
    c->job = c->dset[c->dset_cur].job;
    c->counters_clean = c->dset[c->dset_cur].counters_clean;
    for (auto& e : c->eo) e.valid = false;   // prepared candidates are dropped
    c->seg_staged = false;                   // a staged segment is forgotten (its set is simply staged into again)
    return ICELK_OK;
}

int icelk_seg_detect_finish(icelk_t* h, int max_corners, int* out_n)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    int rc = seg_stage(c, max_corners, out_n);
    if (rc) return rc;
    return seg_switch(c);
}

int icelk_seg_detect(icelk_t* h, int slot, int use_mask, int max_corners, double quality_level, double min_distance,
                     int block_size, int* out_n)
{
    int rc = icelk_seg_detect_begin(h, slot, use_mask, max_corners, quality_level, min_distance, block_size);
    if (rc) return rc;
    return icelk_seg_detect_finish(h, max_corners, out_n);
}

int icelk_seg_track_async(icelk_t* h, int slot_prev, int slot_next, int win_w, int win_h, int max_level, int crit_type,
                          int max_count, double epsilon, double min_eig_threshold, float fb_threshold)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    return seg_track_core(c, slot_prev, slot_next, win_w, win_h, max_level, crit_type, max_count, epsilon,
                          min_eig_threshold, fb_threshold, false);
}

int icelk_seg_track_defer(icelk_t* h, int slot_prev, int slot_next, int win_w, int win_h, int max_level, int crit_type,
                          int max_count, double epsilon, double min_eig_threshold, float fb_threshold)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    return seg_track_core(c, slot_prev, slot_next, win_w, win_h, max_level, crit_type, max_count, epsilon,
                          min_eig_threshold, fb_threshold, true);
}

int icelk_seg_track_len_hint(icelk_t* h, int track_len)
{
    if (!h || track_len < 0) return ICELK_EARG;
    C(h)->track_len_hint = track_len;
    return ICELK_OK;
}

int icelk_seg_template_stats(icelk_t* h, long long* out)
{
    if (!h || !out) return ICELK_EARG;
    out[0] = C(h)->tmpl.taken;
    out[1] = C(h)->tmpl.left;
    return ICELK_OK;
}

int icelk_seg_template_info(icelk_t* h, long long* bytes_per_table, long long* rows, int* state)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    if (bytes_per_table) *bytes_per_table = (long long)c->tmpl.bytes;
    if (rows) *rows = c->tmpl.row_bytes ? (long long)(c->tmpl.bytes / c->tmpl.row_bytes) : 0;
    if (state) *state = c->tmpl.failed ? 2 : (c->tmpl.off ? 1 : 0);
    return ICELK_OK;
}

int icelk_seg_tail_stats(icelk_t* h, long long* out)
{
    if (!h || !out) return ICELK_EARG;
    out[0] = C(h)->tails_dev;
    out[1] = C(h)->tails_host;
    return ICELK_OK;
}

int icelk_seg_flush(icelk_t* h)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    return flush_deferred(c);
}

// run the projection kernel over `n` gathered tracks sitting in d_tracks_in and bring the results to the host
static int project_core(Ctx* c, const float* d_tracks_in, int n, int nv, const icelk_camera_t* cam,
                        const icelk_utm_filter_t* filt, int host_pitch, double* x, double* y, double* u, double* v,
                        double* speed, uint8_t* keep)
{
    const int m = nv - 1;
    const size_t need = (size_t)n * (m > 0 ? m : 1);
    if (need > c->proj_cap) {
        if (c->d_proj) hipFree(c->d_proj);
        c->d_proj = nullptr;
        c->proj_cap = 0;
        int rc = dmalloc(c, &c->d_proj, 5 * need);
        if (!rc && !c->d_keep) rc = dmalloc(c, &c->d_keep, (size_t)c->max_pts);   // n <= max_pts always
        if (rc) return rc;
        c->proj_cap = need;
    }
    double* P[5];
    for (int k = 0; k < 5; k++) P[k] = c->d_proj + (size_t)k * c->proj_cap;
    {
        ProfScope p(c, K_PROJECT);
        launch_project_tracks(c->stream, d_tracks_in, n, nv, *cam, *filt, P[0], P[1], P[2], P[3], P[4], c->d_keep);
    }
    int rc = check_launch(c, "project_tracks");
    if (rc) return rc;
    double* H[5] = {x, y, u, v, speed};
    for (int k = 0; k < 5; k++)
        if (H[k] && m > 0)
            HIPCHK(c, hipMemcpy2DAsync(H[k], sizeof(double) * host_pitch, P[k], sizeof(double) * m, sizeof(double) * m, n,
                                       hipMemcpyDeviceToHost, c->stream));
    if (keep) HIPCHK(c, hipMemcpyAsync(keep, c->d_keep, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ICELK_OK;
}

static int check_projection_args(Ctx* c, const icelk_camera_t* cam, const icelk_utm_filter_t* filt)
{
    if (!cam || !filt) FAIL(c, ICELK_EARG, "camera / filter missing");
    if (!(filt->interval_s > 0)) FAIL(c, ICELK_EARG, "tracking interval must be positive");
    return ICELK_OK;
}

int icelk_project_tracks(icelk_t* h, const float* tracks, int n, int n_vertices, const icelk_camera_t* cam,
                         const icelk_utm_filter_t* filt, double* x, double* y, double* u, double* v, double* speed,
                         uint8_t* keep)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    int rc = check_projection_args(c, cam, filt);
    if (rc) return rc;
    if (n < 0 || n_vertices < 1 || (n > 0 && !tracks)) FAIL(c, ICELK_EARG, "bad track array");
    if (n > c->max_pts) FAIL(c, ICELK_ECAP, "more tracks than max_pts of icelk_create");
    if (n_vertices > kMaxVert) FAIL(c, ICELK_ECAP, "more than 17 vertices per track");
    if (n == 0) return ICELK_OK;
    HIPCHK(c, hipMemcpyAsync(c->d_out_tracks, tracks, sizeof(float) * 2 * (size_t)n * n_vertices, hipMemcpyHostToDevice,
                             c->stream));
    return project_core(c, c->d_out_tracks, n, n_vertices, cam, filt, n_vertices - 1, x, y, u, v, speed, keep);
}

int icelk_seg_project(icelk_t* h, const icelk_camera_t* cam, const icelk_utm_filter_t* filt, int cap, int max_vectors,
                      double* x, double* y, double* u, double* v, double* speed, uint8_t* keep, int* out_n,
                      int* out_vectors)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    int rc = check_projection_args(c, cam, filt);
    if (rc) return rc;
    if (!c->seg_active) FAIL(c, ICELK_ESTATE, "icelk_seg_detect has not been called");
    int n = 0;
    rc = icelk_seg_live(h, &n, nullptr);
    if (rc) return rc;
    Ctx::SegBuf& S = c->sb[c->sb_cur];
    const int nv = S.vert;
    if (out_n) *out_n = n;
    if (out_vectors) *out_vectors = nv - 1;
    if (n > cap || nv - 1 > max_vectors) FAIL(c, ICELK_ECAP, "host buffers too small");
    if (n == 0) return ICELK_OK;
    launch_seg_gather(c->stream, S.alive, S.upper, S.tracks, S.quality, nv, kMaxVert, c->d_out_tracks, c->d_out_quality);
    rc = check_launch(c, "seg_gather");
    if (rc) return rc;
    return project_core(c, c->d_out_tracks, n, nv, cam, filt, max_vectors, x, y, u, v, speed, keep);
}

// ---- gridding of projected velocities (s3_utm_to_gridded_utm.py:391-421) ------------------------

int icelk_points_in_polygon(icelk_t* h, const double* poly_xy, int n_poly, const double* pts_xy, int n_pts,
                            uint8_t* inside)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    if (n_poly < 0 || n_pts < 0 || (n_poly > 0 && !poly_xy) || (n_pts > 0 && (!pts_xy || !inside)))
        FAIL(c, ICELK_EARG, "bad polygon / point arrays");
    if (n_pts == 0) return ICELK_OK;
    DevBufs B;
    double* d_poly = B.get<double>(2 * (size_t)n_poly);
    double* d_pts = B.get<double>(2 * (size_t)n_pts);
    uint8_t* d_out = B.get<uint8_t>((size_t)n_pts);
    if (!d_poly || !d_pts || !d_out) FAIL(c, ICELK_ENOMEM, "hipMalloc failed");
    if (n_poly > 0)
        HIPCHK(c, hipMemcpyAsync(d_poly, poly_xy, sizeof(double) * 2 * n_poly, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_pts, pts_xy, sizeof(double) * 2 * n_pts, hipMemcpyHostToDevice, c->stream));
    launch_points_in_polygon(c->stream, d_poly, n_poly, d_pts, n_pts, d_out);
    int rc = check_launch(c, "points_in_polygon");
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(inside, d_out, n_pts, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ICELK_OK;
}

int icelk_grid_bin(icelk_t* h, const double* x, const double* y, const double* u, const double* v, int n, double left,
                   double top, double spacing, int cols, int rows, const uint8_t* cell_on, int* count, double* mean_u,
                   double* mean_v, double* speed)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    if (n < 0 || cols <= 0 || rows <= 0 || !(spacing > 0) || !cell_on || !count || !mean_u || !mean_v || !speed ||
        (n > 0 && (!x || !y || !u || !v)))
        FAIL(c, ICELK_EARG, "bad gridding arguments");
    if ((long long)cols * rows > (1 << 24) || n > (1 << 27)) FAIL(c, ICELK_ECAP, "grid or point set too large");
    const int ncells = cols * rows;
    // a point lies in one cell, or -- exactly on an edge / corner -- in up to four; 2 n + 1024 keys cover any set
    // whose points are not all on edges, and the count is checked
    const int key_cap = (int)std::min<long long>(2LL * n + 1024, 0x7fffffffLL);
    DevBufs B;
    double* dx = B.get<double>((size_t)n);
    double* dy = B.get<double>((size_t)n);
    double* du = B.get<double>((size_t)n);
    double* dv = B.get<double>((size_t)n);
    uint8_t* d_on = B.get<uint8_t>((size_t)ncells);
    unsigned long long* keys = B.get<unsigned long long>((size_t)key_cap);
    unsigned long long* keys_sorted = B.get<unsigned long long>((size_t)key_cap);
    int* d_key_count = B.get<int>(1);
    int* d_count = B.get<int>((size_t)ncells);
    double* d_mu = B.get<double>((size_t)ncells);
    double* d_mv = B.get<double>((size_t)ncells);
    double* d_sp = B.get<double>((size_t)ncells);
    if (!dx || !dy || !du || !dv || !d_on || !keys || !keys_sorted || !d_key_count || !d_count || !d_mu || !d_mv || !d_sp)
        FAIL(c, ICELK_ENOMEM, "hipMalloc failed");
    const hipStream_t s = c->stream;
    const size_t nb = sizeof(double) * (size_t)n;
    if (n > 0) {
        HIPCHK(c, hipMemcpyAsync(dx, x, nb, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(dy, y, nb, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(du, u, nb, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(dv, v, nb, hipMemcpyHostToDevice, s));
    }
    HIPCHK(c, hipMemcpyAsync(d_on, cell_on, ncells, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemsetAsync(d_key_count, 0, sizeof(int), s));
    launch_grid_assign(s, dx, dy, n, left, top, spacing, cols, rows, d_on, keys, d_key_count, key_cap);
    int rc = check_launch(c, "grid_assign");
    if (rc) return rc;
    int total = 0;
    HIPCHK(c, hipMemcpyAsync(&total, d_key_count, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (total > key_cap) FAIL(c, ICELK_ECAP, "more than two cells per point on average (all points on cell edges?)");
    const unsigned long long* sorted = keys;
    if (total > 1) {
        int cell_bits = 1;
        while ((1 << cell_bits) < ncells) cell_bits++;
        const size_t tmp_bytes = sort_keys_asc(s, nullptr, 0, keys, keys_sorted, total, 32 + cell_bits);
        void* tmp = B.get<uint8_t>(tmp_bytes);
        if (!tmp) FAIL(c, ICELK_ENOMEM, "hipMalloc failed");
        sort_keys_asc(s, tmp, tmp_bytes, keys, keys_sorted, total, 32 + cell_bits);
        rc = check_launch(c, "grid sort");
        if (rc) return rc;
        sorted = keys_sorted;
    }
    launch_grid_reduce(s, sorted, d_key_count, du, dv, ncells, d_count, d_mu, d_mv, d_sp);
    rc = check_launch(c, "grid_reduce");
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(count, d_count, sizeof(int) * ncells, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(mean_u, d_mu, sizeof(double) * ncells, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(mean_v, d_mv, sizeof(double) * ncells, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(speed, d_sp, sizeof(double) * ncells, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return ICELK_OK;
}

// which segment a read-out addresses: the current one, or the one closed by the latest switch
static int seg_pick(Ctx* c, bool closed, int* set)
{
    if (!c->seg_active) FAIL(c, ICELK_ESTATE, "icelk_seg_detect has not been called");
    if (closed && !c->closed_valid) FAIL(c, ICELK_ESTATE, "no closed segment");
    *set = closed ? (c->sb_cur + kSegSets - 1) % kSegSets : c->sb_cur;
    // its waiting pair, if any, belongs to the result
    if (c->defer.pending && c->defer.set == *set) {
        int rc = flush_deferred(c);
        if (rc) return rc;
    }
    return closed ? ICELK_OK : seg_wait(c);
}

static int seg_live_core(Ctx* c, bool closed, int* out_live, int64_t* out_tracked_total)
{
    int set = 0;
    int rc = seg_pick(c, closed, &set);
    if (rc) return rc;
    launch_seg_stats(c->stream, c->sb[set].alive, c->sb[set].upper, c->d_tracked, c->h_seg);
    rc = check_launch(c, "seg_stats");
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int n = (int)c->h_seg[0];
    const unsigned long long t = c->h_seg[1];
    if (out_live) *out_live = n;
    if (out_tracked_total) *out_tracked_total = (int64_t)t;
    return ICELK_OK;
}

int icelk_seg_live(icelk_t* h, int* out_live, int64_t* out_tracked_total)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    return seg_live_core(c, false, out_live, out_tracked_total);
}

int icelk_seg_track(icelk_t* h, int slot_prev, int slot_next, int win_w, int win_h, int max_level, int crit_type,
                    int max_count, double epsilon, double min_eig_threshold, float fb_threshold, int* out_live)
{
    int rc = icelk_seg_track_async(h, slot_prev, slot_next, win_w, win_h, max_level, crit_type, max_count, epsilon,
                                   min_eig_threshold, fb_threshold);
    if (rc) return rc;
    return icelk_seg_live(h, out_live, nullptr);
}

static int seg_read_core(Ctx* c, bool closed, float* tracks, float* quality, int cap, int max_vertices, int* out_n,
                         int* out_vertices)
{
    int n = 0;
    int rc = seg_live_core(c, closed, &n, nullptr);
    if (rc) return rc;
    Ctx::SegBuf& S = c->sb[closed ? (c->sb_cur + kSegSets - 1) % kSegSets : c->sb_cur];
    const int nv = S.vert;
    if (out_n) *out_n = n;
    if (out_vertices) *out_vertices = nv;
    if (!tracks && !quality) return ICELK_OK;
    if (n > cap || nv > max_vertices) FAIL(c, ICELK_ECAP, "host track buffers too small");
    if (n == 0) return ICELK_OK;
    launch_seg_gather(c->stream, S.alive, S.upper, S.tracks, S.quality, nv, kMaxVert, c->d_out_tracks, c->d_out_quality);
    rc = check_launch(c, "seg_gather");
    if (rc) return rc;
    // host layout: (n, max_vertices, 2) and (n, max_vertices-1) with the caller's vertex dimension
    if (tracks)
        HIPCHK(c, hipMemcpy2DAsync(tracks, sizeof(float) * 2 * max_vertices, c->d_out_tracks, sizeof(float) * 2 * nv,
                                   sizeof(float) * 2 * nv, n, hipMemcpyDeviceToHost, c->stream));
    if (quality && nv > 1)
        HIPCHK(c, hipMemcpy2DAsync(quality, sizeof(float) * (max_vertices - 1), c->d_out_quality, sizeof(float) * (nv - 1),
                                   sizeof(float) * (nv - 1), n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ICELK_OK;
}

int icelk_seg_read(icelk_t* h, float* tracks, float* quality, int cap, int max_vertices, int* out_n, int* out_vertices)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    return seg_read_core(c, false, tracks, quality, cap, max_vertices, out_n, out_vertices);
}

int icelk_seg_read_closed(icelk_t* h, float* tracks, float* quality, int cap, int max_vertices, int* out_n,
                          int* out_vertices)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    return seg_read_core(c, true, tracks, quality, cap, max_vertices, out_n, out_vertices);
}

static int seg_archive_core(Ctx* c, bool closed, void* dev_tracks, void* dev_quality, void* dev_count, int cap_rows,
                            int* out_vertices)
{
    if (!dev_tracks || !dev_count) FAIL(c, ICELK_EARG, "null device buffer");
    int set = 0;
    int rc = seg_pick(c, closed, &set);
    if (rc) return rc;
    Ctx::SegBuf& S = c->sb[set];
    if (cap_rows < S.upper) FAIL(c, ICELK_ECAP, "archive rows < tracks of the segment");
    const int nv = S.vert;
    if (out_vertices) *out_vertices = nv;
    // quality is optional: the gather kernel writes it next to the tracks; without a destination it goes to the
    // handle's own read-out buffer
    launch_seg_gather(c->stream, S.alive, S.upper, S.tracks, S.quality, nv, kMaxVert, reinterpret_cast<float*>(dev_tracks),
                      dev_quality ? reinterpret_cast<float*>(dev_quality) : c->d_out_quality, reinterpret_cast<int*>(dev_count));
    rc = check_launch(c, "seg_archive");
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(S.used_own, c->stream));
    S.used = S.used_own;
    return ICELK_OK;
}

int icelk_seg_archive(icelk_t* h, void* dev_tracks, void* dev_quality, void* dev_count, int cap_rows, int* out_vertices)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    return seg_archive_core(c, false, dev_tracks, dev_quality, dev_count, cap_rows, out_vertices);
}

int icelk_seg_archive_closed(icelk_t* h, void* dev_tracks, void* dev_quality, void* dev_count, int cap_rows,
                             int* out_vertices)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    return seg_archive_core(c, true, dev_tracks, dev_quality, dev_count, cap_rows, out_vertices);
}

// ---- measurement -------------------------------------------------------------------------------
int icelk_prof_enable(icelk_t* h, int on)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    if (!on) prof_drain(c);
    c->prof = on != 0;
    c->prof_tracker_only = on == 2;
    return ICELK_OK;
}

int icelk_prof_reset(icelk_t* h)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    prof_drain(c);
    for (int i = 0; i < K_COUNT_; i++) {
        c->prof_launches[i] = 0;
        c->prof_ms[i] = 0;
    }
    return ICELK_OK;
}

int icelk_prof_iterations(icelk_t* h, uint32_t* host_out, int cap, int* out_n)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    HIPCHK(c, hipSetDevice(c->device));
    if (!out_n || cap < 0 || (cap > 0 && !host_out)) FAIL(c, ICELK_EARG, "bad output buffer");
    *out_n = c->iters_n;
    if (host_out && cap > 0 && c->iters_n > 0) {
        const int n = c->iters_n < cap ? c->iters_n : cap;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipMemcpy(host_out, c->d_iters, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost));
    }
    return ICELK_OK;
}

int icelk_stream_probe_info(icelk_t* h, int* picks, double* quickest, double* limit)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    if (picks)
        for (int k = 0; k < 4; k++) picks[k] = c->side_pick[k];
    if (quickest) *quickest = c->probe_quickest;
    if (limit) *limit = c->probe_limit;
    return ICELK_OK;
}

int icelk_prof_count(void) { return K_COUNT_; }

const char* icelk_prof_name(int kernel_id)
{
    if (kernel_id < 0 || kernel_id >= K_COUNT_) return "";
    return kKernelNames[kernel_id];
}

int icelk_prof_get(icelk_t* h, int kernel_id, int* launches, double* total_ms)
{
    if (!h) return ICELK_EARG;
    Ctx* c = C(h);
    if (kernel_id < 0 || kernel_id >= K_COUNT_) FAIL(c, ICELK_EARG, "bad kernel id");
    prof_drain(c);
    if (launches) *launches = c->prof_launches[kernel_id];
    if (total_ms) *total_ms = c->prof_ms[kernel_id];
    return ICELK_OK;
}

}  // extern "C"
