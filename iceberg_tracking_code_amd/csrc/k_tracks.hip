// k_tracks.hip -- device-resident `tracks` / `trackquality` state of the reference's frame loop.
//
// Reference semantics (s1_lucaskanade_tracking.py:335-359): after the forward-backward test only
// tracks with valid == 1 are kept, in their original order; each kept track gets the new vertex
// (x, y) and the distance d appended.  On the GPU a track keeps its row ("origin", its index in
// detection order) in a fixed table and the list of live tracks is a stably compacted array of
// origins, so extending a segment moves 8 + 4 + 4 bytes per surviving track and nothing crosses PCIe.
#include "icelk_internal.h"

namespace icelk {

namespace {

__global__ void k_seg_init(const float* __restrict__ corners, int n, float* __restrict__ live_xy,
                           int* __restrict__ origin, float* __restrict__ tracks, int max_vert, int* n_live,
                           unsigned long long* tracked_total)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *n_live = n;  // tracked_total keeps counting across segments
    if (i >= n) return;
    const float x = corners[2 * i], y = corners[2 * i + 1];
    live_xy[2 * i] = x; live_xy[2 * i + 1] = y;
    origin[i] = i;
    tracks[((size_t)i * max_vert) * 2] = x;
    tracks[((size_t)i * max_vert) * 2 + 1] = y;
}

// Stable compaction by one workgroup (n <= ~10^5; runs in a few microseconds).
__global__ __launch_bounds__(1024) void k_compact(const float* __restrict__ p1, const float* __restrict__ dist,
                                                  const uint8_t* __restrict__ valid,
                                                  const int* __restrict__ origin_in, const int* __restrict__ n_in,
                                                  float* __restrict__ live_out, int* __restrict__ origin_out,
                                                  int* __restrict__ n_out, float* __restrict__ tracks,
                                                  float* __restrict__ quality, int vert, int max_vert,
                                                  unsigned long long* __restrict__ tracked_total)
{
    // tiles of 1024 consecutive tracks: coalesced loads, ballot prefix inside a wave, 16 wave totals in LDS
    __shared__ int wave_tot[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = *n_in;
    int running = 0;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const bool keep = i < n && valid[i];
        const unsigned long long m = __ballot(keep);
        if (lane == 0) wave_tot[wave] = __popcll(m);
        __syncthreads();
        int wbase = 0, tile_total = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int t = wave_tot[k];
            wbase += k < wave ? t : 0;
            tile_total += t;
        }
        if (keep) {
            const int j = running + wbase + __popcll(m & ((1ull << lane) - 1ull));
            const int o = origin_in[i];
            const float x = p1[2 * i], y = p1[2 * i + 1];
            live_out[2 * j] = x; live_out[2 * j + 1] = y;
            origin_out[j] = o;
            tracks[((size_t)o * max_vert + vert) * 2] = x;
            tracks[((size_t)o * max_vert + vert) * 2 + 1] = y;
            quality[(size_t)o * (max_vert - 1) + (vert - 1)] = dist[i];
        }
        running += tile_total;
        __syncthreads();
    }
    if (tid == 0) {
        *n_out = running;
        *tracked_total += (unsigned long long)n;
    }
}

__global__ void k_seg_gather(const int* __restrict__ origin, const int* __restrict__ n_live,
                             const float* __restrict__ tracks, const float* __restrict__ quality, int nvert,
                             int max_vert, float* __restrict__ out_tracks, float* __restrict__ out_quality)
{
    const int n = *n_live;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const int o = origin[j];
        for (int v = 0; v < nvert; v++) {
            out_tracks[((size_t)j * nvert + v) * 2] = tracks[((size_t)o * max_vert + v) * 2];
            out_tracks[((size_t)j * nvert + v) * 2 + 1] = tracks[((size_t)o * max_vert + v) * 2 + 1];
        }
        for (int v = 0; v + 1 < nvert; v++)
            out_quality[(size_t)j * (nvert - 1) + v] = quality[(size_t)o * (max_vert - 1) + v];
    }
}

}  // namespace

void launch_seg_init(hipStream_t s, const float* corners, int n, float* live_xy, int* origin, float* tracks,
                     int max_vert, int* n_live, unsigned long long* tracked_total)
{
    const int blocks = (n > 0 ? n + 255 : 256) / 256;
    hipLaunchKernelGGL(k_seg_init, dim3(blocks), dim3(256), 0, s, corners, n, live_xy, origin, tracks, max_vert,
                       n_live, tracked_total);
}

void launch_compact(hipStream_t s, const float* p1, const float* dist, const uint8_t* valid, const int* origin_in,
                    const int* n_in, float* live_out, int* origin_out, int* n_out, float* tracks, float* quality,
                    int vert, int max_vert, unsigned long long* tracked_total)
{
    hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, s, p1, dist, valid, origin_in, n_in, live_out, origin_out,
                       n_out, tracks, quality, vert, max_vert, tracked_total);
}

void launch_seg_gather(hipStream_t s, const int* origin, const int* n_live, int n_upper, const float* tracks,
                       const float* quality, int nvert, int max_vert, float* out_tracks, float* out_quality)
{
    const int blocks = (n_upper > 0 ? n_upper + 255 : 256) / 256;
    hipLaunchKernelGGL(k_seg_gather, dim3(blocks), dim3(256), 0, s, origin, n_live, tracks, quality, nvert, max_vert,
                       out_tracks, out_quality);
}

}  // namespace icelk
