// k_tracks.hip -- device-resident `tracks` / `trackquality` state of the reference's frame loop.
//
// Reference semantics (s1_lucaskanade_tracking.py:335-359): after the forward-backward test only tracks with
// valid == 1 are kept, in their original order; each kept track gets the new vertex (x, y) and the distance d
// appended.  On the GPU track f keeps row f of a fixed table for the whole segment and an `alive` byte; the
// tracker launch itself appends vertices and clears `alive` (lk_common.h seg_append), so extending a segment
// costs no extra launch and nothing crosses PCIe.  Order-preserving compaction happens once, when a finished
// segment is read out.
#include "lk_common.h"

namespace icelk {

namespace {

__global__ void k_seg_init(const float* __restrict__ corners, int n, float* __restrict__ xy,
                           uint8_t* __restrict__ alive, float* __restrict__ tracks, int max_vert)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = corners[2 * i], y = corners[2 * i + 1];
    xy[2 * i] = x; xy[2 * i + 1] = y;
    alive[i] = 1;
    tracks[((size_t)i * max_vert) * 2] = x;
    tracks[((size_t)i * max_vert) * 2 + 1] = y;
}

// Launch order of a segment's tracks.  Results do not depend on it, time does, twice over:
//  * HBM traffic: the detector hands the corners over in response order, i.e. scattered over the frame, so consecutive
//    workgroups of the tracker touch unrelated cache lines (measured: 5x the algorithmic bytes).  One counting sort per
//    segment bins the tracks into cells of cw x ch px, cells in raster order; the tracker then deals that sequence to the
//    8 XCDs in contiguous eighths (lk_common.h launch_slot), so each XCD's L2 sees one compact part of the frame.
//  * the tail of the launch: a feature whose tiles reach over the frame border is staged through the reflecting loader
//    and masks its template: it takes 1.2-1.6x as long as an interior feature (74 us against 48 us at 4000x3000, 21x21).
//    Launched wherever the sort puts them, some start last and the launch waits for them alone.  They form bin 0 here:
//    the table starts with them, the tracker starts them first (longest-processing-time-first); their number is left in
//    border_count for launch_slot.
constexpr int kOrderBins = 8192;
__global__ __launch_bounds__(1024) void k_seg_order(const float* __restrict__ xy, int n, int cw_shift, int ch_shift,
                                                    int cells_x, int ncells, int w, int h, int border_px,
                                                    int* __restrict__ order, int* __restrict__ border_count)
{
    __shared__ int bins[kOrderBins + 1];
    __shared__ int wave_tot[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = ncells + 1;
    for (int i = tid; i < nb; i += 1024) bins[i] = 0;
    __syncthreads();
    auto cell_of = [&](int i) {
        const int x = (int)xy[2 * i], y = (int)xy[2 * i + 1];
        if (x < border_px || y < border_px || x >= w - border_px || y >= h - border_px) return 0;
        int c = (y >> ch_shift) * cells_x + (x >> cw_shift);
        return 1 + (c < 0 ? 0 : (c >= ncells ? ncells - 1 : c));
    };
    for (int i = tid; i < n; i += 1024) atomicAdd(&bins[cell_of(i)], 1);
    __syncthreads();
    if (tid == 0) *border_count = bins[0];
    // exclusive scan of the bins: 9 consecutive bins per thread, wave prefix by shuffles, 16 wave totals
    int v[9], tsum = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int b = tid * 9 + k;
        v[k] = b < nb ? bins[b] : 0;
        tsum += v[k];
    }
    int incl = tsum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int base = incl - tsum;
    for (int k = 0; k < wave; k++) base += wave_tot[k];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int b = tid * 9 + k;
        if (b < nb) bins[b] = base;
        base += v[k];
    }
    __syncthreads();
    for (int i = tid; i < n; i += 1024) order[atomicAdd(&bins[cell_of(i)], 1)] = i;
}

__global__ __launch_bounds__(1024) void k_seg_stats(const uint8_t* __restrict__ alive, int n,
                                                    const unsigned long long* __restrict__ shards,
                                                    unsigned long long* __restrict__ host_out)
{
    __shared__ int wave_tot[16];
    int cnt = 0;
    for (int i = threadIdx.x; i < n; i += 1024) cnt += alive[i] ? 1 : 0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int k = 0; k < 16; k++) t += wave_tot[k];
        unsigned long long tr = 0;
        for (int k = 0; k < 64; k++) tr += shards[k];
        host_out[0] = (unsigned long long)t;
        host_out[1] = tr;
        __threadfence_system();
    }
}

// One workgroup per tile of 256 tracks, all tiles at once: a workgroup counts the survivors in front of its tile itself
// instead of waiting for the tiles before it (the one-workgroup form of round 1 walked the tiles one after the other: 69 us
// for 10 000 tracks, on the compute stream between two tracker launches).  Round 4: the flags in front of the tile are read
// four at a time (they are 0 / 1 bytes: the dword's bit count is their sum) and the tiles are 256 tracks instead of 1 024 --
// forty workgroups instead of ten for a C2 segment, each with a quarter of the dependent loads: the gather stands on the
// compute stream between the joint launch that closes a segment and the next one (34 us of the 329 us period at C2,
// profiles/r04_c2_timeline.txt).  Inside the tile: ballot prefix inside a wave, 4 wave totals in LDS.  The last tile writes
// the count.
constexpr int kGatherTile = 256;
__global__ __launch_bounds__(kGatherTile) void k_seg_gather(const uint8_t* __restrict__ alive, int n,
                                                            const float* __restrict__ tracks,
                                                            const float* __restrict__ quality, int nvert, int max_vert,
                                                            float* __restrict__ out_tracks, float* __restrict__ out_quality,
                                                            int* __restrict__ out_count)
{
    constexpr int NW = kGatherTile / 64;
    __shared__ int wave_tot[NW];
    __shared__ int wave_before[NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int base = blockIdx.x * kGatherTile;     // a multiple of 4: the flags before it are whole dwords
    int mine = 0;
    const uint32_t* a32 = reinterpret_cast<const uint32_t*>(alive);
    for (int i = tid; i < (base >> 2); i += kGatherTile) mine += __builtin_popcount(a32[i] & 0x01010101u);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mine += __shfl_xor(mine, d, 64);
    const int i = base + tid;
    const bool keep = i < n && alive[i];
    const unsigned long long m = __ballot(keep);
    if (lane == 0) {
        wave_tot[wave] = __popcll(m);
        wave_before[wave] = mine;
    }
    __syncthreads();
    int before = 0, wbase = 0, tile_total = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) {
        const int t = wave_tot[k];
        before += wave_before[k];
        wbase += k < wave ? t : 0;
        tile_total += t;
    }
    if (keep) {
        const int j = before + wbase + __popcll(m & ((1ull << lane) - 1ull));
        const float2* src = reinterpret_cast<const float2*>(tracks) + (size_t)i * max_vert;
        if ((reinterpret_cast<uintptr_t>(out_tracks) & 7u) == 0) {     // the caller's buffer: 8-byte stores where it allows them
            float2* dst = reinterpret_cast<float2*>(out_tracks) + (size_t)j * nvert;
            for (int v = 0; v < nvert; v++) dst[v] = src[v];
        } else {
            float* dst = out_tracks + (size_t)j * nvert * 2;
            for (int v = 0; v < nvert; v++) { const float2 p = src[v]; dst[2 * v] = p.x; dst[2 * v + 1] = p.y; }
        }
        for (int v = 0; v + 1 < nvert; v++)
            out_quality[(size_t)j * (nvert - 1) + v] = quality[(size_t)i * (max_vert - 1) + v];
    }
    if (out_count && tid == 0 && base + kGatherTile >= n) *out_count = before + tile_total;
}

// the forward-backward filter of s1:329-333 on its own: the same device function the fused tracker launches end with
__global__ void k_fb_filter(const float* __restrict__ p0, const float* __restrict__ p0r, int n, float thr, int form,
                            float* __restrict__ dist, uint8_t* __restrict__ valid)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float d = lk::fb_distance(p0[2 * i], p0[2 * i + 1], p0r[2 * i], p0r[2 * i + 1], form);
    dist[i] = d;
    valid[i] = d < thr ? 1 : 0;
}

}  // namespace

void launch_fb_filter(hipStream_t s, const float* p0, const float* p0r, int n, float thr, int form, float* dist,
                      uint8_t* valid)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_fb_filter, dim3((n + 255) / 256), dim3(256), 0, s, p0, p0r, n, thr, form, dist, valid);
}

void launch_seg_init(hipStream_t s, const float* corners, int n, float* xy, uint8_t* alive, float* tracks,
                     int max_vert)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_seg_init, dim3((n + 255) / 256), dim3(256), 0, s, corners, n, xy, alive, tracks, max_vert);
}

void launch_seg_order(hipStream_t s, const float* xy, int n, int w, int h, int border_px, int* order, int* border_count)
{
    if (n <= 0) return;
    int cw = 5, ch = 6;   // 32 x 64 px cells; coarsen until they fit the LDS histogram
    auto cells = [&](int& cx) {
        cx = ((w - 1) >> cw) + 1;
        return cx * (((h - 1) >> ch) + 1);
    };
    int cx = 0;
    while (cells(cx) > kOrderBins) {
        if (cw <= ch) cw++;
        else ch++;
    }
    const int nc = cells(cx);
    hipLaunchKernelGGL(k_seg_order, dim3(1), dim3(1024), 0, s, xy, n, cw, ch, cx, nc, w, h, border_px, order, border_count);
}

void launch_seg_stats(hipStream_t s, const uint8_t* alive, int n, const unsigned long long* tracked_shards,
                      unsigned long long* host_out)
{
    hipLaunchKernelGGL(k_seg_stats, dim3(1), dim3(1024), 0, s, alive, n, tracked_shards, host_out);
}

void launch_seg_gather(hipStream_t s, const uint8_t* alive, int n, const float* tracks, const float* quality,
                       int nvert, int max_vert, float* out_tracks, float* out_quality, int* out_count)
{
    if (n <= 0) {
        if (out_count) hipMemsetAsync(out_count, 0, sizeof(int), s);
        return;
    }
    hipLaunchKernelGGL(k_seg_gather, dim3((n + kGatherTile - 1) / kGatherTile), dim3(kGatherTile), 0, s, alive, n, tracks, quality, nvert, max_vert, out_tracks,
                       out_quality, out_count);
}

}  // namespace icelk
