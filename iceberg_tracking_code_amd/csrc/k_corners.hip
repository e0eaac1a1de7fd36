// k_corners.hip -- Shi-Tomasi corner detection for gfx950.
//
// Replaces cv2.goodFeaturesToTrack(frame_gray, mask=mask, **feature_params) at
// s1_lucaskanade_tracking.py:437 (s0_1_test_lucaskanade_tracking.py:167).  Arithmetic restated
// from OpenCV's cornerMinEigenVal / goodFeaturesToTrack (OpenCV is not part of
// /root/reference).
//
// Stages
//   K6 k_min_eig      fused Sobel -> covariance products -> blockSize^2 box sum -> min eigenvalue,
//                     one tile per workgroup through LDS, plus the masked maximum of the map
//                     (order-preserving atomicMax) -- the f32 Dx, Dy and 3-channel covariance
//                     images OpenCV materialises (5 x 4 B/px) never exist in HBM.
//   K7 k_nms_collect  threshold (max * qualityLevel), 3x3 non-max test, mask, 1-px border;
//                     survivors appended as 64-bit keys (response bits << 32 | raster index).
//   K8 min distance   OpenCV accepts candidates greedily in response order.  Equivalent parallel
//                     form: a candidate is accepted iff no ACCEPTED candidate of higher priority
//                     lies closer than minDistance; iterate "reject if an accepted stronger
//                     neighbour exists / accept if every stronger neighbour is rejected" to the
//                     fixed point over a cell grid of round(minDistance) px (3x3 cell search, as
//                     OpenCV's grid).  Only the accepted set is sorted.
#include "icelk_internal.h"

namespace icelk {

namespace {

__device__ __forceinline__ int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// order-preserving map float -> uint32 (larger float <=> larger key)
__device__ __forceinline__ unsigned ordered_key(float v)
{
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ __forceinline__ float key_to_float(unsigned k)
{
    const unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(b);
#else
    float f;
    memcpy(&f, &b, 4);
    return f;
#endif
}

constexpr int EIG_TW = 64;  // output tile
constexpr int EIG_TH = 16;

// ------------------------------------------------------------------------------------------------
// K6.  LDS: cov[3][(TH+bs-1)][(TW+bs-1)] f32, then hs[3][(TH+bs-1)][TW] f64.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_min_eig(const uint8_t* __restrict__ img, int w, int h, int pitch,
                                                 int bs, float k0, float k1, float* __restrict__ eig,
                                                 const uint8_t* __restrict__ mask, int mask_pitch,
                                                 unsigned* __restrict__ max_key)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int anchor = bs / 2;
    const int ew = EIG_TW + bs - 1, eh = EIG_TH + bs - 1;
    double* hs = reinterpret_cast<double*>(smem);                       // 3 * eh * TW doubles
    float* cov = reinterpret_cast<float*>(smem + sizeof(double) * 3 * eh * EIG_TW);  // 3 * eh * ew floats
    const int x0 = blockIdx.x * EIG_TW, y0 = blockIdx.y * EIG_TH;
    const int tid = threadIdx.x;

    // 1. covariance products at the (reflected) extended positions
    for (int i = tid; i < ew * eh; i += 256) {
        const int ey = i / ew, ex = i - ey * ew;
        const int rx = reflect101(x0 - anchor + ex, w), ry = reflect101(y0 - anchor + ey, h);
        const int xm = reflect101(rx - 1, w), xp = reflect101(rx + 1, w);
        const int ym = reflect101(ry - 1, h), yp = reflect101(ry + 1, h);
        const uint8_t* r0 = img + (size_t)ym * pitch;
        const uint8_t* r1 = img + (size_t)ry * pitch;
        const uint8_t* r2 = img + (size_t)yp * pitch;
        const float a0 = (float)r0[xm], b0 = (float)r0[rx], c0 = (float)r0[xp];
        const float a1 = (float)r1[xm], c1 = (float)r1[xp];
        const float a2 = (float)r2[xm], b2 = (float)r2[rx], c2 = (float)r2[xp];
        // Dx: row pass [-1 0 1] (exact), column pass (r0 + r2)*k1 + r1*k0
        const float dx = __fadd_rn(__fmul_rn(__fadd_rn(__fsub_rn(c0, a0), __fsub_rn(c2, a2)), k1),
                                   __fmul_rn(__fsub_rn(c1, a1), k0));
        // Dy: row pass k1*a + k0*b + k1*c left to right, column pass t2 - t0
        const float t0 = __fadd_rn(__fadd_rn(__fmul_rn(k1, a0), __fmul_rn(k0, b0)), __fmul_rn(k1, c0));
        const float t2 = __fadd_rn(__fadd_rn(__fmul_rn(k1, a2), __fmul_rn(k0, b2)), __fmul_rn(k1, c2));
        const float dy = __fsub_rn(t2, t0);
        cov[i] = __fmul_rn(dx, dx);
        cov[ew * eh + i] = __fmul_rn(dx, dy);
        cov[2 * ew * eh + i] = __fmul_rn(dy, dy);
    }
    __syncthreads();
    // 2. horizontal window sums, left to right, double
    for (int i = tid; i < eh * EIG_TW; i += 256) {
        const int ey = i / EIG_TW, ox = i - ey * EIG_TW;
        const float* c = cov + ey * ew + ox;
        double s0 = 0, s1 = 0, s2 = 0;
        for (int k = 0; k < bs; k++) {
            s0 += (double)c[k];
            s1 += (double)c[ew * eh + k];
            s2 += (double)c[2 * ew * eh + k];
        }
        hs[i] = s0;
        hs[eh * EIG_TW + i] = s1;
        hs[2 * eh * EIG_TW + i] = s2;
    }
    __syncthreads();
    // 3. vertical sums top to bottom, eigenvalue, masked maximum
    unsigned best = 0;  // smaller than the key of any float
    for (int i = tid; i < EIG_TH * EIG_TW; i += 256) {
        const int oy = i / EIG_TW, ox = i - oy * EIG_TW;
        const int x = x0 + ox, y = y0 + oy;
        if (x >= w || y >= h) continue;
        const double* r = hs + oy * EIG_TW + ox;
        double s0 = 0, s1 = 0, s2 = 0;
        for (int k = 0; k < bs; k++) {
            s0 += r[k * EIG_TW];
            s1 += r[eh * EIG_TW + k * EIG_TW];
            s2 += r[2 * eh * EIG_TW + k * EIG_TW];
        }
        const float a = __fmul_rn((float)s0, 0.5f), b = (float)s1, c = __fmul_rn((float)s2, 0.5f);
        const float d = __fsub_rn(a, c);
        const float v = __fsub_rn(__fadd_rn(a, c), sqrtf(__fadd_rn(__fmul_rn(d, d), __fmul_rn(b, b))));
        eig[(size_t)y * w + x] = v;
        if (!mask || mask[(size_t)y * mask_pitch + x]) {
            const unsigned k = ordered_key(v);
            best = k > best ? k : best;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned other = __shfl_xor(best, o);
        best = other > best ? other : best;
    }
    if ((tid & 63) == 0 && best) atomicMax(max_key, best);
}

// ------------------------------------------------------------------------------------------------
// K7.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_nms_collect(const float* __restrict__ eig, int w, int h,
                                                     const uint8_t* __restrict__ mask, int mask_pitch,
                                                     const unsigned* __restrict__ max_key, double quality,
                                                     unsigned long long* __restrict__ cand,
                                                     int* __restrict__ cand_count, int cand_cap)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x + 1;
    const int y = blockIdx.y + 1;
    if (x >= w - 1 || y >= h - 1) return;
    const unsigned mk = *max_key;
    const double max_val = mk ? (double)key_to_float(mk) : 0.0;
    const float thr = (float)(max_val * quality);
    const float v = eig[(size_t)y * w + x];
    if (!(v > thr) || v == 0.f) return;
    if (mask && !mask[(size_t)y * mask_pitch + x]) return;
    float m = 0.f;  // dilate of the TOZERO-thresholded map (includes the centre)
#pragma unroll
    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
        for (int dx = -1; dx <= 1; dx++) {
            float q = eig[(size_t)(y + dy) * w + (x + dx)];
            q = q > thr ? q : 0.f;
            m = q > m ? q : m;
        }
    if (v != m) return;
    const int pos = atomicAdd(cand_count, 1);
    if (pos < cand_cap)
        cand[pos] = ((unsigned long long)ordered_key(v) << 32) | (unsigned)(y * w + x);
}

// ------------------------------------------------------------------------------------------------
// K8 helpers.
// ------------------------------------------------------------------------------------------------
__global__ void k_cell_count(const unsigned long long* __restrict__ cand, const int* __restrict__ n_ptr, int w,
                             int cell, int gw, int* __restrict__ cell_count)
{
    const int n = *n_ptr;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned idx = (unsigned)cand[i];
        const int y = idx / w, x = idx - y * w;
        atomicAdd(&cell_count[(y / cell) * gw + (x / cell)], 1);
    }
}

// single-workgroup exclusive scan: start[i] = sum_{j<i} count[j], start[n] = total
__global__ __launch_bounds__(1024) void k_scan(const int* __restrict__ count, int* __restrict__ start, int n)
{
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int chunk = (n + 1023) / 1024;
    const int lo = tid * chunk, hi = min(n, lo + chunk);
    int s = 0;
    for (int i = lo; i < hi; i++) s += count[i];
    part[tid] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        int v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = tid ? part[tid - 1] : 0;
    for (int i = lo; i < hi; i++) {
        start[i] = run;
        run += count[i];
    }
    if (tid == 1023) start[n] = part[1023];
}

__global__ void k_cell_fill(const unsigned long long* __restrict__ cand, const int* __restrict__ n_ptr, int w,
                            int cell, int gw, const int* __restrict__ cell_start, int* __restrict__ cell_fill,
                            unsigned long long* __restrict__ cell_cand, uint8_t* __restrict__ state)
{
    const int n = *n_ptr;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned long long key = cand[i];
        const unsigned idx = (unsigned)key;
        const int y = idx / w, x = idx - y * w;
        const int c = (y / cell) * gw + (x / cell);
        const int pos = cell_start[c] + atomicAdd(&cell_fill[c], 1);
        cell_cand[pos] = key;
        state[pos] = 0;
    }
}

// One relaxation round.  round_counters[r] counts the candidates still undecided after round r.
__global__ void k_suppress_round(const unsigned long long* __restrict__ cell_cand, const int* __restrict__ n_ptr,
                                 int w, int cell, int gw, int gh, const int* __restrict__ cell_start,
                                 uint8_t* state, double md2, int* __restrict__ round_counters, int r)
{
    if (r > 0 && round_counters[r - 1] == 0) return;
    const int n = *n_ptr;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (state[i]) continue;
        const unsigned long long key = cell_cand[i];
        const unsigned idx = (unsigned)key;
        const int y = idx / w, x = idx - y * w;
        const int xc = x / cell, yc = y / cell;
        const int x1 = max(0, xc - 1), y1 = max(0, yc - 1), x2 = min(gw - 1, xc + 1), y2 = min(gh - 1, yc + 1);
        bool rejected = false, blocked = false;
        for (int yy = y1; yy <= y2 && !rejected; yy++)
            for (int xx = x1; xx <= x2 && !rejected; xx++) {
                const int c = yy * gw + xx;
                const int b = cell_start[c], e = cell_start[c + 1];
                for (int j = b; j < e; j++) {
                    const unsigned long long kj = cell_cand[j];
                    if (kj <= key) continue;  // only stronger candidates matter (keys are unique)
                    const unsigned ij = (unsigned)kj;
                    const int yj = ij / w, xj = ij - yj * w;
                    const float dx = (float)(x - xj), dy = (float)(y - yj);
                    if (!((double)(dx * dx + dy * dy) < md2)) continue;
                    const uint8_t sj = __atomic_load_n(&state[j], __ATOMIC_RELAXED);
                    if (sj == 1) { rejected = true; break; }
                    if (sj == 0) blocked = true;
                }
            }
        if (rejected) __atomic_store_n(&state[i], (uint8_t)2, __ATOMIC_RELAXED);
        else if (!blocked) __atomic_store_n(&state[i], (uint8_t)1, __ATOMIC_RELAXED);
        else atomicAdd(&round_counters[r], 1);
    }
}

__global__ void k_gather_accepted(const unsigned long long* __restrict__ cell_cand, const int* __restrict__ n_ptr,
                                  const uint8_t* __restrict__ state, unsigned long long* __restrict__ acc,
                                  int* __restrict__ acc_count)
{
    const int n = *n_ptr;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (state[i] == 1) acc[atomicAdd(acc_count, 1)] = cell_cand[i];
    }
}

__global__ void k_emit(const unsigned long long* __restrict__ keys, int n, int w, float* __restrict__ xy)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned idx = (unsigned)keys[i];
    const int y = idx / w, x = idx - y * w;
    xy[2 * i] = (float)x;
    xy[2 * i + 1] = (float)y;
}

}  // namespace

void launch_min_eig(hipStream_t s, const Level& img, int block_size, float* eig, const uint8_t* mask,
                    int mask_pitch, unsigned* max_key)
{
    double scale = (double)(1 << 2) * block_size;
    scale *= 255.0;
    scale = 1.0 / scale;
    const float k1 = (float)(1.0 * scale), k0 = (float)(2.0 * scale);
    const int ew = EIG_TW + block_size - 1, eh = EIG_TH + block_size - 1;
    const size_t lds = sizeof(double) * 3 * eh * EIG_TW + sizeof(float) * 3 * eh * ew;
    dim3 grid((img.w + EIG_TW - 1) / EIG_TW, (img.h + EIG_TH - 1) / EIG_TH);
    hipLaunchKernelGGL(k_min_eig, grid, dim3(256), lds, s, img.ptr, img.w, img.h, img.pitch, block_size, k0, k1,
                       eig, mask, mask_pitch, max_key);
}

size_t min_eig_lds_bytes(int block_size)
{
    const int ew = EIG_TW + block_size - 1, eh = EIG_TH + block_size - 1;
    return sizeof(double) * 3 * eh * EIG_TW + sizeof(float) * 3 * eh * ew;
}

void launch_nms_collect(hipStream_t s, const float* eig, int w, int h, const uint8_t* mask, int mask_pitch,
                        const unsigned* max_key, double quality, unsigned long long* cand, int* cand_count,
                        int cand_cap)
{
    if (w < 3 || h < 3) return;
    dim3 grid((w - 2 + 255) / 256, h - 2);
    hipLaunchKernelGGL(k_nms_collect, grid, dim3(256), 0, s, eig, w, h, mask, mask_pitch, max_key, quality, cand,
                       cand_count, cand_cap);
}

// Greedy-equivalent min-distance selection.  D.cand / D.cand_count hold the candidates.  On return
// D.acc / D.acc_count hold the accepted keys (unsorted).  Synchronises the stream (reads the
// round counters).
int run_min_distance(hipStream_t s, DetectScratch& D, int w, int h, int n_cand_upper, double min_distance,
                     std::string& err)
{
    const int cell = (int)lrint(min_distance);
    const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
    const int ncell = gw * gh;
    const double md2 = min_distance * min_distance;
    hipMemsetAsync(D.cell_count, 0, sizeof(int) * (size_t)(ncell + 1), s);
    hipMemsetAsync(D.cell_fill, 0, sizeof(int) * (size_t)ncell, s);
    hipMemsetAsync(D.acc_count, 0, sizeof(int), s);
    const int blocks = max(1, min(2048, (n_cand_upper + 255) / 256));
    hipLaunchKernelGGL(k_cell_count, dim3(blocks), dim3(256), 0, s, D.cand, D.cand_count, w, cell, gw,
                       D.cell_count);
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, s, D.cell_count, D.cell_start, ncell);
    hipLaunchKernelGGL(k_cell_fill, dim3(blocks), dim3(256), 0, s, D.cand, D.cand_count, w, cell, gw, D.cell_start,
                       D.cell_fill, D.cell_cand, D.state);
    constexpr int kBatch = 8;
    int last = 1;
    for (int guard = 0; guard < 4096 && last != 0; guard++) {
        hipMemsetAsync(D.undecided, 0, sizeof(int) * kBatch, s);
        for (int r = 0; r < kBatch; r++)
            hipLaunchKernelGGL(k_suppress_round, dim3(blocks), dim3(256), 0, s, D.cell_cand, D.cand_count, w, cell,
                               gw, gh, D.cell_start, D.state, md2, D.undecided, r);
        int counters[kBatch];
        hipError_t e = hipMemcpyAsync(counters, D.undecided, sizeof(counters), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) { err = hipGetErrorString(e); return ICELK_EHIP; }
        last = counters[kBatch - 1];
        for (int r = 0; r < kBatch; r++)
            if (counters[r] == 0) { last = 0; break; }
    }
    if (last != 0) { err = "min-distance suppression did not converge"; return ICELK_EHIP; }
    hipLaunchKernelGGL(k_gather_accepted, dim3(blocks), dim3(256), 0, s, D.cell_cand, D.cand_count, D.state, D.acc,
                       D.acc_count);
    return ICELK_OK;
}

void launch_emit_corners(hipStream_t s, const unsigned long long* keys, int n, int w, float* xy)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_emit, dim3((n + 255) / 256), dim3(256), 0, s, keys, n, w, xy);
}

}  // namespace icelk
