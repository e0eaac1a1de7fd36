// k_image.hip -- streaming image kernels for gfx950: BGR->gray, pyrDown, procedural frames.
//
// All three are HBM-bound byte streams: one pass, every source byte fetched
// from HBM once, wide coalesced accesses, no LDS (the 5x5 pyrDown footprint of a thread is kept in
// registers as a rolling window of horizontal sums; neighbouring lanes overlap by a few bytes that
// are served by the CU's vector L1).
//
// Arithmetic restated from OpenCV (not in /root/reference; see oracle/icelk_oracle.c header):
//   gray     : cv2.cvtColor(frame, cv2.COLOR_BGR2GRAY)           s1_lucaskanade_tracking.py:311
//   pyrDown  : inside cv2.calcOpticalFlowPyrLK                   s1_lucaskanade_tracking.py:323,326
#include "icelk_internal.h"

namespace icelk {

__device__ __forceinline__ int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// ------------------------------------------------------------------------------------------------
// K1  BGR -> gray.  4 pixels per lane: 12 source bytes (3 dwords), 1 dword stored.
// gray = (c0*k0 + c1*k1 + c2*k2 + half) >> shift      
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bgr2gray(const uint8_t* __restrict__ src, int src_pitch,
                                                  uint8_t* __restrict__ dst, int dst_pitch, int w, int h,
                                                  int k0, int k1, int k2, int shift, int aligned)
{
    const int half = 1 << (shift - 1);
    const int quads = (w + 3) >> 2;
    for (int y = blockIdx.y; y < h; y += gridDim.y) {
        const uint8_t* s = src + (size_t)y * src_pitch;
        uint8_t* d = dst + (size_t)y * dst_pitch;
        for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += gridDim.x * blockDim.x) {
            const int x = q << 2;
            if (aligned && x + 4 <= w) {
                const uint32_t* s32 = reinterpret_cast<const uint32_t*>(s + 3 * x);
                uint32_t a = s32[0], b = s32[1], c = s32[2];
                // bytes: a = B0 G0 R0 B1 | b = G1 R1 B2 G2 | c = R2 B3 G3 R3
                int g0 = ((a & 255) * k0 + ((a >> 8) & 255) * k1 + ((a >> 16) & 255) * k2 + half) >> shift;
                int g1 = ((a >> 24) * k0 + (b & 255) * k1 + ((b >> 8) & 255) * k2 + half) >> shift;
                int g2 = (((b >> 16) & 255) * k0 + (b >> 24) * k1 + (c & 255) * k2 + half) >> shift;
                int g3 = (((c >> 8) & 255) * k0 + ((c >> 16) & 255) * k1 + (c >> 24) * k2 + half) >> shift;
                // write-through (a relaxed system-scope store: sc0 sc1), so that the 12 MB leave the L2s as they are written and
                // not at the write-back that ends the kernel
                __hip_atomic_store(reinterpret_cast<uint32_t*>(d + x),
                                   (uint32_t)g0 | ((uint32_t)g1 << 8) | ((uint32_t)g2 << 16) | ((uint32_t)g3 << 24), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
            } else {
                for (int i = x; i < w && i < x + 4; i++)
                    d[i] = (uint8_t)((s[3 * i] * k0 + s[3 * i + 1] * k1 + s[3 * i + 2] * k2 + half) >> shift);
            }
        }
    }
}

void launch_bgr2gray(hipStream_t s, const uint8_t* src, int src_pitch, uint8_t* dst, int dst_pitch, int w,
                     int h, int variant)
{
    int k0, k1, k2, sh;
    if (variant == ICELK_GRAY_CV4) { k0 = 3735; k1 = 19235; k2 = 9798; sh = 15; }
    else { k0 = 1868; k1 = 9617; k2 = 4899; sh = 14; }
    int aligned = (((uintptr_t)src | (uintptr_t)src_pitch | (uintptr_t)dst | (uintptr_t)dst_pitch) & 3) == 0;
    int quads = (w + 3) >> 2;
    dim3 block(256);
    dim3 grid((quads + 255) / 256, h < 4096 ? h : 4096);
    hipLaunchKernelGGL(k_bgr2gray, grid, block, 0, s, src, src_pitch, dst, dst_pitch, w, h, k0, k1, k2, sh,
                       aligned);
}

// ------------------------------------------------------------------------------------------------
// K2  pyrDown 8-bit.  dst(x,y) = (sum_{i,j} k[i]k[j] src(2x-2+i, 2y-2+j) + 128) >> 8, k=[1 4 6 4 1],
//     reflect-101 at the borders.
//
// A lane owns a strip of 4 output columns x PD_ROWS output rows.  Per source row it needs the 11
// bytes at columns 8q-2 .. 8q+8; it loads the 16 aligned bytes 8q-4 .. 8q+11 as one dwordx4
// (neighbouring lanes overlap by 8 bytes -> L1 hits, HBM sees every byte once) and forms the four
// horizontal sums; the 2*PD_ROWS+3 row loads of a strip are all in flight together.  One output
// row is 4 bytes = 1 dword per lane, 256 B per wave.
// ------------------------------------------------------------------------------------------------
constexpr int PD_ROWS = 4;
constexpr int PD_SRC = 2 * PD_ROWS + 3;  // source rows a lane touches

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

struct HSum4 {
    int v[4];
};

__device__ __forceinline__ HSum4 hsum_from_bytes(const int* b)  // b[0..10] = columns 8q-2 .. 8q+8
{
    HSum4 r;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int* c = b + 2 * i;  // c[0..4] = columns 2x-2 .. 2x+2
        r.v[i] = c[0] + c[4] + 4 * (c[1] + c[3]) + 6 * c[2];
    }
    return r;
}

__device__ __forceinline__ HSum4 hsum_from_u4(const u32x4_a4 u)  // the 16 bytes at columns 8q-4 .. 8q+11
{
    int b[11];
    b[0] = (u.x >> 16) & 255; b[1] = u.x >> 24;
    b[2] = u.y & 255; b[3] = (u.y >> 8) & 255; b[4] = (u.y >> 16) & 255; b[5] = u.y >> 24;
    b[6] = u.z & 255; b[7] = (u.z >> 8) & 255; b[8] = (u.z >> 16) & 255; b[9] = u.z >> 24;
    b[10] = u.w & 255;
    return hsum_from_bytes(b);
}

__device__ __forceinline__ HSum4 hsum_border(const uint8_t* row, int q, int w)
{
    int b[11];
#pragma unroll
    for (int i = 0; i < 11; i++) b[i] = row[reflect101(8 * q - 2 + i, w)];
    return hsum_from_bytes(b);
}

__global__ __launch_bounds__(64) void k_pyrdown(const uint8_t* __restrict__ src, int sw, int sh, int spitch,
                                                uint8_t* __restrict__ dst, int dw, int dh, int dpitch)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;  // group of 4 output columns
    const int oy0 = blockIdx.y * PD_ROWS;
    if (4 * q >= dw || oy0 >= dh) return;
    // the 16-byte load covers columns 8q-4 .. 8q+11: fast path only when fully inside the row
    const bool fast = (q > 0) && (8 * q + 12 <= sw);

    // all source rows of the strip are requested before the first one is used (latency, not issue, bound)
    HSum4 r[PD_SRC];
    if (fast) {
        u32x4_a4 u[PD_SRC];
#pragma unroll
        for (int k = 0; k < PD_SRC; k++) {
            const uint8_t* row = src + (size_t)reflect101(2 * oy0 - 2 + k, sh) * spitch;
            u[k] = *reinterpret_cast<const u32x4_a4*>(row + 8 * q - 4);
        }
#pragma unroll
        for (int k = 0; k < PD_SRC; k++) r[k] = hsum_from_u4(u[k]);
    } else {
#pragma unroll
        for (int k = 0; k < PD_SRC; k++) {
            const uint8_t* row = src + (size_t)reflect101(2 * oy0 - 2 + k, sh) * spitch;
            r[k] = hsum_border(row, q, sw);
        }
    }
#pragma unroll
    for (int j = 0; j < PD_ROWS; j++) {
        const int oy = oy0 + j;
        if (oy < dh) {
            uint32_t out = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                int v = r[2 * j].v[i] + r[2 * j + 4].v[i] + 4 * (r[2 * j + 1].v[i] + r[2 * j + 3].v[i]) + 6 * r[2 * j + 2].v[i];
                out |= (uint32_t)((v + 128) >> 8) << (8 * i);
            }
            uint8_t* d = dst + (size_t)oy * dpitch + 4 * q;
            if (4 * q + 4 <= dw) {
                *reinterpret_cast<uint32_t*>(d) = out;
            } else {
                for (int i = 0; 4 * q + i < dw; i++) d[i] = (uint8_t)(out >> (8 * i));
            }
        }
    }
}

void launch_pyrdown(hipStream_t s, const Level& src, const Level& dst)
{
    const int quads = (dst.w + 3) / 4;
    dim3 block(64, 1);
    // 64 lanes wide so that a wave stores 256 contiguous bytes per output row
    dim3 grid((quads + 63) / 64, (dst.h + PD_ROWS - 1) / PD_ROWS);
    hipLaunchKernelGGL(k_pyrdown, grid, block, 0, s, src.ptr, src.w, src.h, src.pitch, dst.ptr, dst.w, dst.h,
                       dst.pitch);
}

// ------------------------------------------------------------------------------------------------
// Procedural frame (bit-identical to iceberg_tracking_code_amd/synth.py::frame).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t hash2(uint32_t ix, uint32_t iy, uint32_t seedmul)
{
    uint32_t h = (ix * 0x9E3779B1u) ^ (iy * 0x85EBCA77u) ^ seedmul;
    h ^= h >> 15; h *= 0x2C1B3C6Du;
    h ^= h >> 12; h *= 0x297A2D39u;
    h ^= h >> 15;
    return h;
}

__device__ __forceinline__ uint32_t octave(uint32_t X, uint32_t Y, int k, uint32_t seed)
{
    const uint32_t sh = 8 + k;
    const uint32_t cx = X >> sh, cy = Y >> sh;
    const uint32_t fx = (X >> k) & 255u, fy = (Y >> k) & 255u;
    const uint32_t sx = (fx * fx * (768u - 2u * fx)) >> 16;
    const uint32_t sy = (fy * fy * (768u - 2u * fy)) >> 16;
    const uint32_t sm = (uint32_t)(((uint64_t)(seed + 7919u * (uint32_t)k) * 0xC2B2AE3Dull) & 0xFFFFFFFFull);
    const uint32_t v00 = hash2(cx, cy, sm) >> 24, v10 = hash2(cx + 1, cy, sm) >> 24;
    const uint32_t v01 = hash2(cx, cy + 1, sm) >> 24, v11 = hash2(cx + 1, cy + 1, sm) >> 24;
    const uint32_t top = v00 * (256u - sx) + v10 * sx;
    const uint32_t bot = v01 * (256u - sx) + v11 * sx;
    return (top * (256u - sy) + bot * sy) >> 16;
}

// The frame is the texture sampled at  X = x0 + 256 x + ((ax x + bx y) >> 12),  Y = y0 + 256 y + ((ay x + by y) >> 12)
// (1/256 px; ax.. in 2^-20 px per px): a translation plus, optionally, a small affine deformation (shear / scale /
// rotation of a fraction of a percent), so that the motion differs from place to place in the frame.
__global__ __launch_bounds__(256) void k_synth(uint8_t* __restrict__ dst, int w, int h, int pitch, uint32_t x0,
                                               uint32_t y0, uint32_t seed, int ax, int bx, int ay, int by)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (4 * q >= w || y >= h) return;
    uint32_t out = 0;
    for (int i = 0; i < 4; i++) {
        const int x = 4 * q + i;
        const uint32_t X = x0 + ((uint32_t)x << 8) + (uint32_t)((ax * x + bx * y) >> 12);
        const uint32_t Y = y0 + ((uint32_t)y << 8) + (uint32_t)((ay * x + by * y) >> 12);
        uint32_t acc = 3u * octave(X, Y, 3, seed) + 3u * octave(X, Y, 5, seed) + 2u * octave(X, Y, 2, seed);
        out |= ((acc >> 3) & 255u) << (8 * i);
    }
    uint8_t* d = dst + (size_t)y * pitch + 4 * q;
    if (4 * q + 4 <= w) *reinterpret_cast<uint32_t*>(d) = out;
    else
        for (int i = 0; 4 * q + i < w; i++) d[i] = (uint8_t)(out >> (8 * i));
}

void launch_synth(hipStream_t s, const Level& dst, int64_t ux, int64_t uy, uint32_t seed, const int* affine)
{
    const int64_t bias = (int64_t)1 << 16;
    const uint32_t x0 = (uint32_t)((bias << 8) + ux);
    const uint32_t y0 = (uint32_t)((bias << 8) + uy);
    dim3 block(256);
    dim3 grid(((dst.w + 3) / 4 + 255) / 256, dst.h);
    hipLaunchKernelGGL(k_synth, grid, block, 0, s, dst.ptr, dst.w, dst.h, dst.pitch, x0, y0, seed, affine ? affine[0] : 0,
                       affine ? affine[1] : 0, affine ? affine[2] : 0, affine ? affine[3] : 0);
}

}  // namespace icelk
