// k_tail.hip -- the tail of a detection driven by the device-side counts (no host round trip).
//
// Replaces s1_lucaskanade_tracking.py:437-448 for a detection that starts a segment: the part of
// cv2.goodFeaturesToTrack behind the min-distance rule -- std::sort(tmpCorners, greaterThanPtr()), the maxCorners cut,
// the (x, y) list -- and the reset `tracks = [[(x, y)] for x, y in p.reshape(-1, 2)]` of s1:440-448.
//
// Until round 3 that part needed the host: the number of accepted corners went to pinned memory, the host thread -- which
// also issues the uploads and the tracker launches -- picked it up, and enqueued a library radix sort (5-7 launches), the
// corner list, the new segment's tables and their launch order behind it: a dozen dependent launches that each had to find
// room beside a tracker launch, and a host that stood inside the chain.  Here the chain stays on the device, in four short
// launches of ONE-WAVE workgroups (a single retiring tracker wave makes room for one) that read the counts where the
// min-distance stage left them.  No sort is run: a corner's place in the list is its RANK, and the rank of a key is
//       (keys in the response bins above its own) + (keys of its own bin that are above it)
// for ANY monotone binning of the response -- here 4096 equal slices of the key range [threshold, maximum] of this frame
// (keys are unique: response key << 32 | y << 16 | x, so the order is total and the result is the sorted list bit for bit):
//   k_tail_gather   the accepted candidates -> `acc` (as k_gather_accepted) + a histogram of their response bins; the
//                   workgroup that finishes last scans it into bin offsets and leaves the verdict in `ctl`
//   k_tail_scatter  every key to a slot of its bin (`sorted`: grouped by bin, unordered inside)
//   k_tail_rank     rank = bin offset + keys of the bin above it.  A key ranked below maxCorners IS corner `rank`: its
//                   (x, y) goes straight into the new segment's tables (position, alive flag, vertex 0) and into the
//                   histogram of the launch order; the workgroup that finishes last turns that into offsets
//   k_tail_order    the launch order of the new segment (what k_seg_order does: border features first, then cells in
//                   raster order), the reset of the detector set's counters for its next detection, and -- by the
//                   workgroup that finishes last -- the counts for the host: {candidates, accepted, undecided, pruned,
//                   corners, status} + the sequence word the host polls.
// The host reads the counts when it adopts the segment (icelk_seg_detect_stage / _switch): nothing waits for it.  When the
// device-side verdict is "not valid" -- the relaxation of the min-distance stage has not converged, a pruned candidate
// set yielded fewer than maxCorners corners (k_corners.hip), the corners exceed the tables, or one response bin holds
// more than 65536 keys (a frame of identical corners: ranking inside it is quadratic) -- nothing is written and the host
// runs the tail of before (detect_finish), which is also what ICELK_HOST_TAIL=1 selects.
#include "icelk_internal.h"

namespace icelk {

namespace {

constexpr int CT = 64;
constexpr int NB = kTailRespBins;
constexpr int kMaxBinKeys = 65536;

__device__ __forceinline__ unsigned ordered_key(float v)
{
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(unsigned k)
{
    const unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(b);
}

// Monotone map response key -> bin, bin 0 = strongest: NB equal slices of [lowest key a kept candidate can have, maximum]
struct RespBins {
    unsigned max_key;
    int shift;
};
__device__ __forceinline__ RespBins resp_bins(const unsigned* max_key_p, const unsigned* prune_key_p, double quality)
{
    RespBins r;
    r.max_key = *max_key_p;
    const double max_val = r.max_key ? (double)key_to_float(r.max_key) : 0.0;
    unsigned lo = ordered_key((float)(max_val * quality));
    const unsigned pk = *prune_key_p;
    lo = pk > lo ? pk : lo;
    const unsigned span = r.max_key > lo ? r.max_key - lo : 0u;
    r.shift = 0;
    while ((span >> r.shift) >= (unsigned)NB) r.shift++;
    return r;
}
__device__ __forceinline__ int resp_bin(const RespBins& r, unsigned long long key)
{
    const unsigned hk = (unsigned)(key >> 32);
    if (hk >= r.max_key) return 0;
    const unsigned b = (r.max_key - hk) >> r.shift;
    return b >= (unsigned)NB ? NB - 1 : (int)b;
}

// inclusive prefix sum over the 64 lanes through the DPP network (no LDS round trips)
__device__ __forceinline__ int wave_incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);  // row_bcast15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);  // row_bcast31 -> rows 2, 3
    return v;
}

// exclusive scan of cnt[0 .. nb) by ONE wave into out_a (and out_b, if given); rows of 64 consecutive bins, read coalesced,
// sixteen rows in flight at a time (a row at a time the wave paid the memory latency nb / 64 times in a row: 130 us for
// 8192 bins beside a tracker launch).  The counts were formed by device-scope atomics of other workgroups: read likewise.
// Returns the largest count (every lane); *first_out = the count of bin 0.
__device__ __forceinline__ int scan_one_wave(const int* cnt, int nb, int* out_a, int* out_b, int lane, int* first_out = nullptr)
{
    constexpr int ROWS = 16;
    int run = 0, biggest = 0;
    for (int r0 = 0; r0 < nb; r0 += CT * ROWS) {
        int v[ROWS];
#pragma unroll
        for (int k = 0; k < ROWS; k++) {
            const int b = r0 + k * CT + lane;
            v[k] = b < nb ? __hip_atomic_load(&cnt[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        }
        if (first_out && r0 == 0) *first_out = __builtin_amdgcn_readlane(v[0], 0);
#pragma unroll
        for (int k = 0; k < ROWS; k++) {
            const int b = r0 + k * CT + lane;
            const int incl = wave_incl_scan(v[k]);
            if (b < nb) {
                out_a[b] = run + incl - v[k];
                if (out_b) out_b[b] = run + incl - v[k];
            }
            run += __builtin_amdgcn_readlane(incl, CT - 1);
            biggest = v[k] > biggest ? v[k] : biggest;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const int t = __shfl_xor(biggest, o);
        biggest = t > biggest ? t : biggest;
    }
    return biggest;
}

__device__ __forceinline__ int order_cell(float fx, float fy, const TailOrderGeo& g)
{
    const int x = (int)fx, y = (int)fy;
    if (x < g.border_px || y < g.border_px || x >= g.w - g.border_px || y >= g.h - g.border_px) return 0;
    const int c = (y >> g.ch_shift) * g.cells_x + (x >> g.cw_shift);
    return 1 + (c < 0 ? 0 : (c >= g.ncells ? g.ncells - 1 : c));
}

// Is this workgroup the last of the grid to get here?  What the last one goes on to read of the others' work are words they
// updated with device-scope atomics (histograms, counters): those are performed at the point of coherence, so it is enough
// that a wave's own atomics have been acknowledged before it draws its ticket (a wait, no cache write-back -- a device-scope
// fence in each of a thousand workgroups writes back the L2 of an XCD on which a tracker launch is leaving its templates).
__device__ __forceinline__ bool last_arriver(int* ticket, int lane)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    int last = 0;
    if (lane == 0) last = atomicAdd(ticket, 1) == (int)gridDim.x - 1 ? 1 : 0;
    last = __builtin_amdgcn_readfirstlane(last);
    return last != 0;
}

struct TailArgs {
    const unsigned* max_key;
    const unsigned* prune_key;
    double quality;
    int* ctl;
    int* hist;     // [NB] keys per response bin
    int* start;    // [NB + 1] first slot of the bin in `sorted`
    int* cursor;   // [NB] next free slot of the bin (k_tail_scatter); afterwards: end of the bin
};

__global__ __launch_bounds__(CT) void k_tail_gather(const unsigned long long* __restrict__ cell_cand,
                                                     const int* __restrict__ n_ptr, const uint8_t* __restrict__ state,
                                                     unsigned long long* __restrict__ acc, int* acc_count, TailArgs A,
                                                     const int* __restrict__ undecided_p, int max_corners, int cap,
                                                     int force_status)
{
    const int lane = threadIdx.x;
    const int n = *n_ptr;
    const RespBins rb = resp_bins(A.max_key, A.prune_key, A.quality);
    for (int i0 = (int)blockIdx.x * CT; i0 < n; i0 += (int)gridDim.x * CT) {
        const int i = i0 + lane;
        const bool keep = i < n && state[i] == 1;
        const unsigned long long key = keep ? cell_cand[i] : 0ull;
        const unsigned long long m = __ballot(keep);
        if (m == 0ull) continue;
        int base = 0;
        if (lane == 0) base = atomicAdd(acc_count, __popcll(m));
        base = __shfl(base, 0);
        if (keep) {
            acc[base + __popcll(m & ((1ull << lane) - 1ull))] = key;
            atomicAdd(&A.hist[resp_bin(rb, key)], 1);
        }
    }
    if (!last_arriver(&A.ctl[TC_TICKET_GATHER], lane)) return;
    // the last workgroup: bin offsets and the verdict
    const int total = __hip_atomic_load(acc_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int undec = *undecided_p;
    const bool pruned = *A.prune_key != 0u;
    const int n_out = (max_corners > 0 && total > max_corners) ? max_corners : total;
    const int biggest = scan_one_wave(A.hist, NB, A.start, A.cursor, lane);
    int status = TAIL_OK;
    if (undec != 0) status = TAIL_UNDECIDED;
    else if (pruned && max_corners > 0 && total < max_corners) status = TAIL_PRUNED_SHORT;
    else if (n_out > cap) status = TAIL_OVERFLOW;
    else if (biggest > kMaxBinKeys) status = TAIL_SKEWED;
    if (force_status) status = force_status;   // tests: the host's tail behind a device verdict (ICELK_TAIL_FORCE_STATUS)
    if (lane == 0) {
        A.start[NB] = total;
        A.ctl[TC_NOUT] = status == TAIL_OK ? n_out : 0;
        A.ctl[TC_STATUS] = status;
        A.ctl[TC_CAND] = n;
        A.ctl[TC_ACC] = total;
        A.ctl[TC_UNDECIDED] = undec;
        A.ctl[TC_PRUNED] = pruned ? 1 : 0;
    }
}

__global__ __launch_bounds__(CT) void k_tail_scatter(const unsigned long long* __restrict__ acc,
                                                      unsigned long long* __restrict__ sorted, TailArgs A)
{
    if (A.ctl[TC_STATUS] != TAIL_OK) return;
    const int n = A.ctl[TC_ACC];
    const RespBins rb = resp_bins(A.max_key, A.prune_key, A.quality);
    for (int i = (int)blockIdx.x * CT + (int)threadIdx.x; i < n; i += (int)gridDim.x * CT) {
        const unsigned long long key = acc[i];
        sorted[atomicAdd(&A.cursor[resp_bin(rb, key)], 1)] = key;
    }
}

__global__ __launch_bounds__(CT) void k_tail_rank(const unsigned long long* __restrict__ sorted, TailArgs A,
                                                   float* __restrict__ seg_xy, uint8_t* __restrict__ seg_alive,
                                                   float* __restrict__ seg_tracks, int max_vert, TailOrderGeo geo, int* bins,
                                                   int* __restrict__ border_count)
{
    const int lane = threadIdx.x;
    const int status = A.ctl[TC_STATUS];
    if (status == TAIL_OK) {
        const int n = A.ctl[TC_ACC], n_out = A.ctl[TC_NOUT];
        const RespBins rb = resp_bins(A.max_key, A.prune_key, A.quality);
        for (int i = (int)blockIdx.x * CT + lane; i < n; i += (int)gridDim.x * CT) {
            const unsigned long long key = sorted[i];
            const int b = resp_bin(rb, key);
            const int lo = A.start[b], hi = A.start[b + 1];
            if (lo >= n_out) continue;   // every key of this bin ranks behind the cut
            int above = lo;
            for (int q = lo; q < hi; q++) above += sorted[q] > key ? 1 : 0;
            if (above < n_out) {
                const unsigned idx = (unsigned)key;
                const float x = (float)(int)(idx & 0xffffu), y = (float)(int)(idx >> 16);
                seg_xy[2 * above] = x;
                seg_xy[2 * above + 1] = y;
                seg_alive[above] = 1;
                seg_tracks[((size_t)above * max_vert) * 2] = x;
                seg_tracks[((size_t)above * max_vert) * 2 + 1] = y;
                if (bins) atomicAdd(&bins[order_cell(x, y, geo)], 1);
            }
        }
    }
    if (!last_arriver(&A.ctl[TC_TICKET_RANK], lane)) return;
    if (status == TAIL_OK && bins) {
        int nborder = 0;   // bin 0 = the border features
        scan_one_wave(bins, geo.ncells + 1, bins, nullptr, lane, &nborder);
        if (lane == 0) *border_count = nborder;
    }
}

__global__ __launch_bounds__(CT) void k_tail_order(TailArgs A, const float* __restrict__ seg_xy, TailOrderGeo geo, int* bins,
                                                    int* __restrict__ order, int order_blocks, TailReset rs,
                                                    int* __restrict__ host_counts, int counts_seq_word, int seq)
{
    const int lane = threadIdx.x;
    int* ctl = A.ctl;
    const int status = ctl[TC_STATUS], n_out = ctl[TC_NOUT];
    if ((int)blockIdx.x < order_blocks) {
        if (status == TAIL_OK && order)
            for (int i = (int)blockIdx.x * CT + lane; i < n_out; i += order_blocks * CT)
                order[atomicAdd(&bins[order_cell(seg_xy[2 * i], seg_xy[2 * i + 1], geo)], 1)] = i;
    } else {
        const int i0 = ((int)blockIdx.x - order_blocks) * CT + lane, stride = ((int)gridDim.x - order_blocks) * CT;
        for (int i = i0; i < NB; i += stride) A.hist[i] = 0;    // this file's own histogram: always
        if (status == TAIL_OK) {
            // the detector set's counters for its next detection (what k_detect_reset does); not when the host is going to
            // run the tail itself: it needs them
            for (int i = i0; i <= rs.ncell; i += stride) {
                rs.cell_count[i] = 0;
                if (i < rs.ncell) rs.cell_fill[i] = 0;
                if (i % rs.scan_chunk == 0) rs.chunk_tot[(i / rs.scan_chunk) * rs.chunk_stride] = 0;
            }
            for (int i = i0; i < rs.key_bins; i += stride) rs.key_hist[i] = 0;
            if (i0 < 8) rs.undecided[i0] = 0;
            if (i0 == 0) {
                *rs.acc_count = 0;
                *rs.cand_count = 0;
                *rs.prune_key = 0;
            }
        }
    }
    if (!last_arriver(&ctl[TC_TICKET_ORDER], lane)) return;
    if (bins)
        for (int b = lane; b <= geo.ncells + 1; b += CT) bins[b] = 0;   // for this set's next detection
    if (lane == 0) {
        host_counts[0] = ctl[TC_CAND];
        host_counts[1] = ctl[TC_ACC];
        host_counts[2] = ctl[TC_UNDECIDED];
        host_counts[3] = ctl[TC_PRUNED];
        host_counts[4] = n_out;
        host_counts[5] = status;
        ctl[TC_TICKET_GATHER] = 0;
        ctl[TC_TICKET_RANK] = 0;
        ctl[TC_TICKET_ORDER] = 0;
        __threadfence_system();
        __hip_atomic_store(host_counts + counts_seq_word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

TailArgs args_of(DetectScratch& D, double quality)
{
    TailArgs A{};
    A.max_key = D.max_key;
    A.prune_key = D.prune_key;
    A.quality = quality;
    A.ctl = D.tail_ctl;
    A.hist = D.tail_resp;
    A.start = D.tail_resp + (NB + 64);
    A.cursor = D.tail_resp + 2 * (NB + 64);
    return A;
}

}  // namespace

TailOrderGeo tail_order_geometry(int w, int h, int border_px)
{
    TailOrderGeo g{};
    int cw = 5, ch = 6;   // 32 x 64 px cells, coarsened until they fit the bins (as launch_seg_order)
    auto cells = [&](int& cx) {
        cx = ((w - 1) >> cw) + 1;
        return cx * (((h - 1) >> ch) + 1);
    };
    int cx = 0;
    while (cells(cx) > kTailOrderBins) {
        if (cw <= ch) cw++;
        else ch++;
    }
    g.ncells = cells(cx);
    g.cells_x = cx;
    g.cw_shift = cw;
    g.ch_shift = ch;
    g.w = w;
    g.h = h;
    g.border_px = border_px;
    return g;
}

size_t tail_resp_words() { return 3 * (size_t)(NB + 64); }

void launch_tail_gather(hipStream_t s, DetectScratch& D, int ncell, double quality, int undecided_index, int max_corners,
                        int cap, int force_status)
{
    hipLaunchKernelGGL(k_tail_gather, dim3(256), dim3(CT), 0, s, D.cell_cand, D.cell_start + ncell, D.state, D.acc,
                       D.acc_count, args_of(D, quality), D.undecided + undecided_index, max_corners, cap, force_status);
}

void launch_tail_device(hipStream_t s, DetectScratch& D, double quality, float* seg_xy, uint8_t* seg_alive, float* seg_tracks,
                        int max_vert, int* order, int* order_border, const TailOrderGeo& geo, const TailReset& rs,
                        int* host_counts, int counts_seq_word, int seq)
{
    int* bins = order ? D.tail_bins : nullptr;
    const TailArgs A = args_of(D, quality);
    hipLaunchKernelGGL(k_tail_scatter, dim3(256), dim3(CT), 0, s, D.acc, D.acc_sorted, A);
    hipLaunchKernelGGL(k_tail_rank, dim3(256), dim3(CT), 0, s, D.acc_sorted, A, seg_xy, seg_alive, seg_tracks, max_vert, geo, bins,
                       order_border);
    // few workgroups: each draws a ticket from ONE word (~90 atomics per us), and each has to find a wave slot
    const int order_blocks = 64;
    int reset_blocks = (rs.ncell + 8 * CT) / (8 * CT);
    if (reset_blocks < 16) reset_blocks = 16;
    if (reset_blocks > 192) reset_blocks = 192;
    hipLaunchKernelGGL(k_tail_order, dim3(order_blocks + reset_blocks), dim3(CT), 0, s, A, seg_xy, geo, bins, order, order_blocks, rs,
                       host_counts, counts_seq_word, seq);
}

}  // namespace icelk
