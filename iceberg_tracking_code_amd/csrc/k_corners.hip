// k_corners.hip -- Shi-Tomasi corner detection for gfx950.
//
// Replaces cv2.goodFeaturesToTrack(frame_gray, mask=mask, **feature_params) at
// s1_lucaskanade_tracking.py:437 (s0_1_test_lucaskanade_tracking.py:167).  Arithmetic restated from OpenCV's
// cornerMinEigenVal / goodFeaturesToTrack (OpenCV is not part of /root/reference).
//
// Stages
//   K6+K7 fused (k_eig_nms<BS>, blockSize 3/5/7/10): one tile per workgroup, everything through LDS:
//         u8 tile -> Sobel -> covariance products -> blockSize^2 box sums (ordered double sums, register
//         blocked) -> min eigenvalue (+1 px halo) -> 3x3 non-max test, mask, 1-px border -> local maxima
//         appended as 64-bit keys (response key << 32 | y << 16 | x), plus the masked maximum of the map
//         (order-preserving atomicMax).  Neither OpenCV's f32 Dx/Dy/covariance images (5 x 4 B/px) nor the
//         eigenvalue map itself ever exist in HBM: 1 B/px is read, ~8 B per local maximum written.
//         The quality threshold (max * qualityLevel) is applied to the candidate list afterwards
//         (k_filter): for max > 0,  v > thr && v == dilate3x3(threshold_tozero(eig))  <=>  v > thr && v >= its
//         8 raw neighbours; for max <= 0 OpenCV finds no corner either.
//   K6, K7 separate (k_min_eig, k_nms_collect): any other blockSize, and the eigenvalue-map read-back.
//   K8 min distance: OpenCV accepts candidates greedily in response order.  Equivalent parallel form: a
//         candidate is accepted iff no ACCEPTED candidate of higher priority lies closer than minDistance;
//         relax "reject if an accepted stronger neighbour exists / accept if every stronger neighbour is
//         rejected" to the fixed point over a cell grid of round(minDistance) px (3x3 cell search, as OpenCV's
//         grid).  Only the accepted set is sorted (k_sort.hip).
// Candidates live in per-workgroup regions (region b = the local maxima of tile b, count in blk_count[b]):
// no single-address atomics anywhere on the image-sized passes (one word saturates at ~90 atomics/us on
// gfx950), and the masked maximum is published with a read-guarded atomicMax.
// The whole detection is enqueued without host round trips; the host synchronises once, to learn the
// number of accepted corners.
#include <cstdlib>

#include "icelk_internal.h"

namespace icelk {

namespace {

__device__ __forceinline__ int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// order-preserving map float -> uint32 (larger float <=> larger key); 0 is below every float
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

// low half of a candidate key: (y << 16) | x orders exactly like the raster address y*w + x (the tie-break of
// OpenCV's greaterThanPtr) and needs no division to unpack; frames are < 65536 px on either side
__device__ __forceinline__ unsigned pack_xy(int x, int y) { return ((unsigned)y << 16) | (unsigned)x; }

__device__ __forceinline__ float threshold_of(const unsigned* max_key, double quality)
{
    const unsigned mk = *max_key;
    const double max_val = mk ? (double)key_to_float(mk) : 0.0;
    return (float)(max_val * quality);
}

// Sobel (ksize 3) pair scaled as cornerEigenValsVecs does, in OpenCV's operation order:
//   Dx: row pass [-1 0 1] (exact), column pass (r0 + r2)*k1 + r1*k0
//   Dy: row pass k1*a + k0*b + k1*c left to right, column pass t2 - t0
// `variant` (icelk_set_variant, the generic kernel only): bit 0 = the symmetric column pass fused, fmaf(r0 + r2, k1, r1 k0)
// (OpenCV 4.x SymmColumnSmallVec_32f in an FMA3 build); bit 1 = the row pass with its two additions fused (a RowFilter loop
// contracted by a compiler that targets FMA); bit 2 = calcMinEigenVal's (a-c)^2 + b^2 as fmaf(b, b, t t).  0 = what every
// other kernel computes.  oracle/icelk_oracle.c names the same switches (orc_set_variant).
__device__ __forceinline__ float row_smooth(float a, float b, float c, float k0, float k1, int variant)
{
    if (variant & 2) return fmaf(k1, c, fmaf(k0, b, __fmul_rn(k1, a)));
    return __fadd_rn(__fadd_rn(__fmul_rn(k1, a), __fmul_rn(k0, b)), __fmul_rn(k1, c));
}

__device__ __forceinline__ void sobel_cov(float a0, float b0, float c0, float a1, float c1, float a2, float b2,
                                          float c2, float k0, float k1, float& xx, float& xy, float& yy, int variant = 0)
{
    const float dx = (variant & 1) ? fmaf(__fadd_rn(__fsub_rn(c0, a0), __fsub_rn(c2, a2)), k1, __fmul_rn(__fsub_rn(c1, a1), k0))
                                   : __fadd_rn(__fmul_rn(__fadd_rn(__fsub_rn(c0, a0), __fsub_rn(c2, a2)), k1),
                                               __fmul_rn(__fsub_rn(c1, a1), k0));
    const float t0 = row_smooth(a0, b0, c0, k0, k1, variant);
    const float t2 = row_smooth(a2, b2, c2, k0, k1, variant);
    const float dy = __fsub_rn(t2, t0);
    xx = __fmul_rn(dx, dx);
    xy = __fmul_rn(dx, dy);
    yy = __fmul_rn(dy, dy);
}

// the same for two horizontally adjacent positions at once, on packed f32 math (v_pk_add/mul_f32: two lanes of
// work per issue slot); every operation rounds exactly like its scalar counterpart
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void sobel_cov2(f2 a0, f2 b0, f2 c0, f2 a1, f2 c1, f2 a2, f2 b2, f2 c2, float k0, float k1,
                                           f2& xx, f2& xy, f2& yy)
{
    const f2 dx = ((c0 - a0) + (c2 - a2)) * k1 + (c1 - a1) * k0;
    const f2 t0 = (a0 * k1 + b0 * k0) + c0 * k1;
    const f2 t2 = (a2 * k1 + b2 * k0) + c2 * k1;
    const f2 dy = t2 - t0;
    xx = dx * dx;
    xy = dx * dy;
    yy = dy * dy;
}

// N consecutive window sums of BS terms each over v[0 .. N+BS-2].  Every term is a float product of derivative
// values of an 8-bit image: magnitudes in [2^-27, 2^-3] with 24-bit mantissas, so ANY sum of up to a few hundred of
// them is exact in double (tests/test_oracle_kat.py::test_box_sums_of_the_covariance_planes_are_exact_in_double) --
// a sliding sum (drop the oldest term, add the next) therefore gives bit for bit the value of OpenCV's running
// RowSum / ColumnSum and of the term-by-term sum the oracle forms, with 2 additions per output instead of BS - 1.
template <int N, int BS>
__device__ __forceinline__ void window_sums(const double (&v)[N + BS - 1], double (&out)[N])
{
    if constexpr (BS <= 3) {
#pragma unroll
        for (int i = 0; i < N; i++) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < BS; k++) s += v[i + k];
            out[i] = s;
        }
    } else {
        double s = v[0];
#pragma unroll
        for (int k = 1; k < BS; k++) s += v[k];
        out[0] = s;
#pragma unroll
        for (int i = 1; i < N; i++) {
            s = (s - v[i - 1]) + v[i + BS - 1];
            out[i] = s;
        }
    }
}

__device__ __forceinline__ float min_eig_of(double s0, double s1, double s2, int variant = 0)
{
    const float a = __fmul_rn((float)s0, 0.5f), b = (float)s1, c = __fmul_rn((float)s2, 0.5f);
    const float d = __fsub_rn(a, c);
    const float r = (variant & 4) ? fmaf(b, b, __fmul_rn(d, d)) : __fadd_rn(__fmul_rn(d, d), __fmul_rn(b, b));
    return __fsub_rn(__fadd_rn(a, c), sqrtf(r));
}

// wave-reduce `best` and publish it; the atomic is skipped when the published maximum is already as large
// (the value only grows, so a stale read can only cause a redundant atomic, never a lost one)
__device__ __forceinline__ void publish_max(unsigned* max_key, unsigned best, int tid)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned other = __shfl_xor(best, o);
        best = other > best ? other : best;
    }
    if ((tid & 63) == 0 && best > __atomic_load_n(max_key, __ATOMIC_RELAXED)) atomicMax(max_key, best);
}

// Candidate source: nblk regions of `region` keys each, blk_count[b] valid keys in region b.
struct CandSrc {
    const unsigned long long* keys;
    const int* blk_count;
    int nblk;
    int region;
};

// ------------------------------------------------------------------------------------------------
// Fused K6+K7, compile-time blockSize: 33 KB of LDS, 4 workgroups per CU.  Only the two derivative planes are kept
// (the covariance products are formed where they are summed, each with the single f32 rounding OpenCV's stored
// planes have), and the double row sums of ONE plane at a time: row pass -> barrier -> column pass into registers
// -> barrier, three times.  (A first form that kept all three covariance planes and their row sums -- 70 KB of LDS,
// 2 workgroups per CU -- took 337 us alone at 12 MP against 120 us for this one, and beside the tracker launch its
// workgroups waited for whole CUs to drain: 440 us against 177 us.  It is gone.)
// ------------------------------------------------------------------------------------------------
template <int BS>
struct EigCfg {
    static constexpr int TW = 64, TH = 16;
    static constexpr int EW = TW + 2, EH = TH + 2;
    static constexpr int CW = EW + BS - 1, CH = EH + BS - 1;
    static constexpr int CWP = (CW + 3) & ~3;
    static constexpr int UW = CW + 2, UH = CH + 2;
    static constexpr int UPD = (UW + 2) / 4 + 1;
    static constexpr int AN = BS / 2;
    static constexpr int RX = 6, RY = 6;
    static constexpr int NGX = EW / RX, NGY = EH / RY;
    static_assert(EW % RX == 0 && EH % RY == 0, "blocking must divide the tile");
    static_assert(EW * NGY <= 256, "one column task per thread");
    static constexpr int U_BYTES = (UPD * UH * 4 + 15) & ~15;
    static constexpr int D_BYTES = 2 * CH * CWP * 4;          // dx, dy
    static constexpr int A_BYTES = U_BYTES + D_BYTES;         // later reused for the eigenvalue tile
    static constexpr int HS_BYTES = CH * EW * 8;              // row sums of one plane
    static constexpr int LDS_BYTES = A_BYTES + HS_BYTES;
    static_assert(EW * EH * 4 <= A_BYTES, "eigenvalue tile must fit the dead derivative region");
};

__device__ __forceinline__ void sobel_d(float a0, float b0, float c0, float a1, float c1, float a2, float b2, float c2,
                                        float k0, float k1, float& dx, float& dy)
{
    dx = __fadd_rn(__fmul_rn(__fadd_rn(__fsub_rn(c0, a0), __fsub_rn(c2, a2)), k1), __fmul_rn(__fsub_rn(c1, a1), k0));
    const float t0 = __fadd_rn(__fadd_rn(__fmul_rn(k1, a0), __fmul_rn(k0, b0)), __fmul_rn(k1, c0));
    const float t2 = __fadd_rn(__fadd_rn(__fmul_rn(k1, a2), __fmul_rn(k0, b2)), __fmul_rn(k1, c2));
    dy = __fsub_rn(t2, t0);
}

__device__ __forceinline__ void sobel_d2(f2 a0, f2 b0, f2 c0, f2 a1, f2 c1, f2 a2, f2 b2, f2 c2, float k0, float k1,
                                         f2& dx, f2& dy)
{
    dx = ((c0 - a0) + (c2 - a2)) * k1 + (c1 - a1) * k0;
    const f2 t0 = (a0 * k1 + b0 * k0) + c0 * k1;
    const f2 t2 = (a2 * k1 + b2 * k0) + c2 * k1;
    dy = t2 - t0;
}

template <int BS>
__global__ __launch_bounds__(256) void k_eig_nms(const uint8_t* __restrict__ img, int w, int h, int pitch, float k0,
                                                  float k1, const uint8_t* __restrict__ mask, int mask_pitch,
                                                  unsigned* __restrict__ max_key,
                                                  unsigned long long* __restrict__ raw, int* __restrict__ blk_count,
                                                  float* __restrict__ eig_out)
{
    using C = EigCfg<BS>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t* U = reinterpret_cast<uint32_t*>(smem);
    float* D = reinterpret_cast<float*>(smem + C::U_BYTES);        // [2][CH][CWP]: dx, dy
    double* hs = reinterpret_cast<double*>(smem + C::A_BYTES);     // [CH][EW]
    float* E = reinterpret_cast<float*>(smem);                     // after the last row pass
    __shared__ int s_list_n;
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * C::TW, y0 = blockIdx.y * C::TH;
    const int ux0 = x0 - 2 - C::AN, uy0 = y0 - 2 - C::AN;
    const bool interior = ux0 >= 0 && uy0 >= 0 && ux0 + C::UW <= w && uy0 + C::UH <= h;
    constexpr int DP = C::CH * C::CWP;   // plane stride

    if (interior) {
        const uint8_t* base = img + (size_t)uy0 * pitch + (ux0 & ~3);
        for (int i = tid; i < C::UPD * C::UH; i += 256) {
            const int r = i / C::UPD, c = i - r * C::UPD;
            U[i] = *reinterpret_cast<const uint32_t*>(base + (size_t)r * pitch + 4 * c);
        }
        __syncthreads();
        const int cs = ux0 & 3;
        constexpr int NQF = C::CW / 4, REM = C::CW - 4 * NQF;
        for (int t = tid; t < C::CH * NQF; t += 256) {
            const int cy = t / NQF, q = t - cy * NQF;
            float F[3][6];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const uint32_t* p = U + (cy + r) * C::UPD + ((cs + 4 * q) >> 2);
                const int sh = (cs + 4 * q) & 3;
                const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
                const uint32_t e0 = __builtin_amdgcn_alignbyte(d1, d0, sh), e1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
                F[r][0] = (float)(e0 & 255); F[r][1] = (float)((e0 >> 8) & 255); F[r][2] = (float)((e0 >> 16) & 255);
                F[r][3] = (float)(e0 >> 24); F[r][4] = (float)(e1 & 255); F[r][5] = (float)((e1 >> 8) & 255);
            }
            f2 dx[2], dy[2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int j = 2 * i;
                sobel_d2(f2{F[0][j], F[0][j + 1]}, f2{F[0][j + 1], F[0][j + 2]}, f2{F[0][j + 2], F[0][j + 3]},
                         f2{F[1][j], F[1][j + 1]}, f2{F[1][j + 2], F[1][j + 3]},
                         f2{F[2][j], F[2][j + 1]}, f2{F[2][j + 1], F[2][j + 2]}, f2{F[2][j + 2], F[2][j + 3]},
                         k0, k1, dx[i], dy[i]);
            }
            float* c = D + cy * C::CWP + 4 * q;
            *reinterpret_cast<float4*>(c) = make_float4(dx[0].x, dx[0].y, dx[1].x, dx[1].y);
            *reinterpret_cast<float4*>(c + DP) = make_float4(dy[0].x, dy[0].y, dy[1].x, dy[1].y);
        }
        if (REM > 0) {
            const uint8_t* Ub = reinterpret_cast<const uint8_t*>(U);
            for (int i = tid; i < C::CH * REM; i += 256) {
                const int cy = i / REM, cx = 4 * NQF + (i - cy * REM);
                const uint8_t* r0 = Ub + cy * C::UPD * 4 + cs + cx;
                const uint8_t* r1 = r0 + C::UPD * 4;
                const uint8_t* r2 = r1 + C::UPD * 4;
                float dx, dy;
                sobel_d((float)r0[0], (float)r0[1], (float)r0[2], (float)r1[0], (float)r1[2], (float)r2[0], (float)r2[1],
                        (float)r2[2], k0, k1, dx, dy);
                D[cy * C::CWP + cx] = dx;
                D[DP + cy * C::CWP + cx] = dy;
            }
        }
    } else {
        for (int i = tid; i < C::CH * C::CW; i += 256) {
            const int cy = i / C::CW, cx = i - cy * C::CW;
            const int rx = reflect101(ux0 + 1 + cx, w), ry = reflect101(uy0 + 1 + cy, h);
            const int xm = reflect101(rx - 1, w), xp = reflect101(rx + 1, w);
            const int ym = reflect101(ry - 1, h), yp = reflect101(ry + 1, h);
            const uint8_t* r0 = img + (size_t)ym * pitch;
            const uint8_t* r1 = img + (size_t)ry * pitch;
            const uint8_t* r2 = img + (size_t)yp * pitch;
            float dx, dy;
            sobel_d((float)r0[xm], (float)r0[rx], (float)r0[xp], (float)r1[xm], (float)r1[xp], (float)r2[xm],
                    (float)r2[rx], (float)r2[xp], k0, k1, dx, dy);
            D[cy * C::CWP + cx] = dx;
            D[DP + cy * C::CWP + cx] = dy;
        }
    }
    __syncthreads();

    // three planes in turn: xx = dx*dx, xy = dx*dy, yy = dy*dy (one f32 rounding each, as the stored planes of the
    // first form had); row sums left to right into hs, column sums top to bottom into registers
    const bool col_task = tid < C::EW * C::NGY;
    const int cg = tid / C::EW, cex = tid - cg * C::EW;   // column task: group of RY rows, column
    double S[3][C::RY];
#pragma unroll
    for (int p = 0; p < 3; p++) {
        const float* A = D + (p == 2 ? DP : 0);
        const float* Bp = D + (p == 0 ? 0 : DP);
        for (int t = tid; t < C::CH * C::NGX; t += 256) {
            const int cy = t / C::NGX, g = t - cy * C::NGX;
            const float* a = A + cy * C::CWP + g * C::RX;
            const float* b = Bp + cy * C::CWP + g * C::RX;
            double v[C::RX + BS - 1];
#pragma unroll
            for (int i = 0; i < C::RX + BS - 1; i++) v[i] = (double)__fmul_rn(a[i], b[i]);
            double* o = hs + cy * C::EW + g * C::RX;
            double r[C::RX];
            window_sums<C::RX, BS>(v, r);
#pragma unroll
            for (int i = 0; i < C::RX; i++) o[i] = r[i];
        }
        __syncthreads();
        if (col_task) {
            double v[C::RY + BS - 1];
#pragma unroll
            for (int i = 0; i < C::RY + BS - 1; i++) v[i] = hs[(cg * C::RY + i) * C::EW + cex];
            window_sums<C::RY, BS>(v, S[p]);
        }
        __syncthreads();   // hs is rewritten by the next plane; after the last one D is dead as well
    }

    unsigned best = 0;
    if (col_task) {
#pragma unroll
        for (int i = 0; i < C::RY; i++) {
            const float e = min_eig_of(S[0][i], S[1][i], S[2][i]);
            const int ey = cg * C::RY + i;
            E[ey * C::EW + cex] = e;
            const int x = x0 - 1 + cex, y = y0 - 1 + ey;
            if (cex >= 1 && cex <= C::TW && ey >= 1 && ey <= C::TH && x < w && y < h) {
                if (eig_out) eig_out[(size_t)y * w + x] = e;
                if (!mask || mask[(size_t)y * mask_pitch + x]) {
                    const unsigned k = ordered_key(e);
                    best = k > best ? k : best;
                }
            }
        }
    }
    publish_max(max_key, best, tid);
    if (tid == 0) s_list_n = 0;
    __syncthreads();

    const int bid = blockIdx.y * gridDim.x + blockIdx.x;
    unsigned long long* region = raw + (size_t)bid * (C::TW * C::TH);
    for (int i = tid; i < C::TW * C::TH; i += 256) {
        const int oy = i / C::TW, ox = i - oy * C::TW;
        const int x = x0 + ox, y = y0 + oy;
        if (x < 1 || y < 1 || x >= w - 1 || y >= h - 1) continue;
        const float* e = E + (oy + 1) * C::EW + (ox + 1);
        const float v = e[0];
        if (!(v > 0.f)) continue;
        float m = e[-C::EW - 1];
        m = fmaxf(m, e[-C::EW]); m = fmaxf(m, e[-C::EW + 1]);
        m = fmaxf(m, e[-1]); m = fmaxf(m, e[1]);
        m = fmaxf(m, e[C::EW - 1]); m = fmaxf(m, e[C::EW]); m = fmaxf(m, e[C::EW + 1]);
        if (v < m) continue;
        if (mask && !mask[(size_t)y * mask_pitch + x]) continue;
        region[atomicAdd(&s_list_n, 1)] = ((unsigned long long)ordered_key(v) << 32) | pack_xy(x, y);
    }
    __syncthreads();
    if (tid == 0) blk_count[bid] = s_list_n;
}

// ------------------------------------------------------------------------------------------------
// K6+K7 in STRIPS (round 3; the default for blockSize 3/5/7/10).  k_eig_nms pays for its 64x16 tile twice: the Sobel stage
// runs on 27x75 positions and the row sums on 27x66 for 16x64 outputs (1.98x / 1.74x), and its column pass and eigenvalue
// stage occupy 198 of 256 threads.  Here a workgroup walks DOWN a strip 245 outputs wide (blockSize 10) and 62 high:
//   * one thread per covariance column.  The Sobel pair rolls down the column in registers (row difference and row
//     smooth of the two pixel rows above are kept; three byte loads per new row, issued one row ahead; BORDER_REFLECT_101
//     in x costs nothing: the three column offsets of a thread are fixed).  The COLUMN sums come first: a running double
//     sum per plane, + the new product, - the one blockSize rows up, kept as f32 in a register ring (static indices: the
//     row loop is unrolled by lcm(blockSize, 4)).  All of these sums are exact (window_sums' note), so summing columns
//     first and sliding the window give bit for bit the value of OpenCV's row-then-column running sums.
//   * every 4 rows: the column sums of the 4 rows (LDS, 24 KB) -> row sums by threads that own 4 consecutive outputs of
//     one row (sliding window, 15 additions per plane) -> min eigenvalue -> an 8-row ring of the eigenvalue map (LDS);
//     then the 3x3 non-max test, one thread per column, for the 4 rows whose lower neighbour now exists.
// Strips that touch the top or bottom of the image (FRESH) reflect rows: the covariance row at a reflected position is
// the one AT that position with its own neighbours, so the roll does not apply there and each row loads its 3x3 afresh.
// Halo: 1.045 in x, 69 covariance rows for 58 output rows.  Regions: 4 per workgroup (16 output rows each).
// Measured and dropped: a form that stays within the 72 VGPRs the tracker's five waves leave free on a SIMD (ring in
// LDS, the running sums and the roll parked in LDS during the row-sum phase, s_setprio 3), so that its workgroups are
// resident the moment they are dispatched instead of waiting for tracker waves to retire on all four SIMDs of a CU.
// It does what it was built for -- the gaps between tracker launches shrink from 11-200 us to 13-20 us -- but the
// tracker launch beside it takes 318 us instead of 282 and the kernel alone 154 us instead of 75 (53 KB of LDS, two
// workgroups per CU): 5 630-5 690 pairs/s against 5 910-5 920 on the same box.  The pipeline is bound by the
// instructions the SIMDs issue, not by when they issue them.
// ------------------------------------------------------------------------------------------------
constexpr int gcd_c(int a, int b) { return b == 0 ? a : gcd_c(b, a % b); }

// NT_ = 256: four waves, one row of a batch each in the row-sum phase.  NT_ = 64 (round 4): ONE wave per workgroup -- the
// strip is 53 outputs wide at blockSize 10 (halo 1.21 in x instead of 1.045), the four rows of a batch are row-summed by
// 4 x 14 lanes at once, and no barrier ever waits for another wave.  What it is for: a four-wave workgroup of 128 VGPRs
// finds room beside a tracker launch only where tracker waves retire on all four SIMDs of a CU at the same moment -- it
// took 200-580 us there for 75 us of work, and since the tail of a detection no longer waits for the host (k_tail.hip)
// that kernel paced the whole C2 pipeline -- whereas a one-wave workgroup takes the place of ANY single retiring wave.
template <int BS, int NT_ = 256>
struct StripCfg {
    static constexpr int NT = NT_;                 // threads = covariance columns
    static constexpr int AN = BS / 2;
    static constexpr int EW = NT - (BS - 1);       // eigenvalue columns
    static constexpr int TW = EW - 2;              // output columns
    static constexpr int R = 4;                    // rows per batch = waves
    static constexpr int UNROLL = BS * R / gcd_c(BS, R);   // rows per trip of the row loop: ring slots and batch rows static
    // eigenvalue rows: whole trips -- an exit from inside the unrolled trip would meet the register ring in a different
    // rotation at every batch (measured: 230 VGPRs instead of 146)
    static constexpr int EH = (64 / UNROLL) * UNROLL;
    static constexpr int SH = EH - 2;              // output rows
    static constexpr int NROWS = EH + BS - 1;      // covariance rows walked
    static constexpr int RX = 4;                   // outputs per row task
    static constexpr int NG = (EW + RX - 1) / RX;  // row tasks per row
    static constexpr int VP = NT + 2;              // doubles per row of column sums (the last task reads 2 beyond)
    static constexpr int SUB = 16, NSUB = (SH + SUB - 1) / SUB;
    static constexpr int V_BYTES = R * 3 * VP * 8;
    static constexpr int E_BYTES = 8 * NT * 4;
    static constexpr int LDS_BYTES = V_BYTES + E_BYTES;
    static_assert(EH % R == 0 && NG <= 64 && RX * NG <= NT + RX - 1, "one wave per row of a batch");
    static_assert(NT == 256 || (NT == 64 && R * NG <= 64), "one-wave form: the rows of a batch side by side in the wave");
    static_assert(RX * (NG - 1) + RX + BS - 2 < VP, "row tasks stay inside a row of column sums");
    static_assert((VP * 8) % 16 == 0, "16-byte reads of the column sums");
};

struct SobelRoll {
    float d1, d2, t1, t2;   // row difference / row smooth of pixel rows yc (1) and yc - 1 (2)
};

__device__ __forceinline__ float smooth3(float a, float b, float c, float k0, float k1)
{
    return __fadd_rn(__fadd_rn(__fmul_rn(k1, a), __fmul_rn(k0, b)), __fmul_rn(k1, c));
}

template <int BS, bool FRESH, int NT_ = 256>
__device__ __forceinline__ void strip_body(const uint8_t* __restrict__ img, int w, int h, int pitch, float k0, float k1,
                                           const uint8_t* __restrict__ mask, int mask_pitch, unsigned* __restrict__ max_key,
                                           unsigned long long* __restrict__ raw, int* __restrict__ blk_count,
                                           float* __restrict__ eig_out, uint8_t* smem, int* s_cnt)
{
    using C = StripCfg<BS, NT_>;
    double* Vb = reinterpret_cast<double*>(smem);                 // [R][3][VP]
    float* Er = reinterpret_cast<float*>(smem + C::V_BYTES);      // [8][NT]
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * C::TW, y0 = blockIdx.y * C::SH;
    const int xs = x0 - 1 - C::AN, ys = y0 - 1 - C::AN;           // image position of covariance column / row 0
    const int rx = reflect101(xs + tid, w);
    const int xm = reflect101(rx - 1, w), xp = reflect101(rx + 1, w);
    const int bid = blockIdx.y * gridDim.x + blockIdx.x;
    if (tid < C::NSUB) s_cnt[tid] = 0;

    float ring[3][BS];
#pragma unroll
    for (int k = 0; k < BS; k++) ring[0][k] = ring[1][k] = ring[2][k] = 0.f;
    double V[3] = {0.0, 0.0, 0.0};
    SobelRoll S{};
    unsigned pa = 0, pb = 0, pc = 0;   // the pixel row one ahead of the roll
    if (!FRESH) {
        const uint8_t* r2 = img + (size_t)(ys - 1) * pitch;
        const uint8_t* r1 = r2 + pitch;
        const uint8_t* r0 = r1 + pitch;
        const float a2 = (float)r2[xm], b2 = (float)r2[rx], c2 = (float)r2[xp];
        const float a1 = (float)r1[xm], b1 = (float)r1[rx], c1 = (float)r1[xp];
        S.d2 = __fsub_rn(c2, a2); S.t2 = smooth3(a2, b2, c2, k0, k1);
        S.d1 = __fsub_rn(c1, a1); S.t1 = smooth3(a1, b1, c1, k0, k1);
        pa = r0[xm]; pb = r0[rx]; pc = r0[xp];
    }
    // products of covariance row r (image row ys + r)
    auto cov_row = [&](int r, float& xx, float& xy, float& yy) {
        float dtop, dmid, dbot, ttop, tbot;
        if (FRESH) {
            const int ry = reflect101(ys + r, h);
            const uint8_t* q0 = img + (size_t)reflect101(ry - 1, h) * pitch;
            const uint8_t* q1 = img + (size_t)ry * pitch;
            const uint8_t* q2 = img + (size_t)reflect101(ry + 1, h) * pitch;
            const float a0 = (float)q0[xm], b0 = (float)q0[rx], c0 = (float)q0[xp];
            const float a1 = (float)q1[xm], c1 = (float)q1[xp];
            const float a2 = (float)q2[xm], b2 = (float)q2[rx], c2 = (float)q2[xp];
            dtop = __fsub_rn(c0, a0); dmid = __fsub_rn(c1, a1); dbot = __fsub_rn(c2, a2);
            ttop = smooth3(a0, b0, c0, k0, k1); tbot = smooth3(a2, b2, c2, k0, k1);
        } else {
            const float a = (float)pa, b = (float)pb, c = (float)pc;
            int nr = ys + r + 2;                  // next pixel row; past the last row needed it is only kept inside the image
            nr = nr < h ? nr : h - 1;
            const uint8_t* q = img + (size_t)nr * pitch;
            pa = q[xm]; pb = q[rx]; pc = q[xp];
            dtop = S.d2; dmid = S.d1; dbot = __fsub_rn(c, a);
            ttop = S.t2; tbot = smooth3(a, b, c, k0, k1);
            S.d2 = S.d1; S.d1 = dbot; S.t2 = S.t1; S.t1 = tbot;
        }
        const float dx = __fadd_rn(__fmul_rn(__fadd_rn(dtop, dbot), k1), __fmul_rn(dmid, k0));
        const float dy = __fsub_rn(tbot, ttop);
        xx = __fmul_rn(dx, dx); xy = __fmul_rn(dx, dy); yy = __fmul_rn(dy, dy);
    };

    // the first blockSize - 1 covariance rows only fill the window
#pragma unroll
    for (int r = 0; r < BS - 1; r++) {
        float p[3];
        cov_row(r, p[0], p[1], p[2]);
#pragma unroll
        for (int q = 0; q < 3; q++) { V[q] += (double)p[q]; ring[q][r] = p[q]; }
    }

    unsigned best = 0;
    // row-sum phase: which row of the batch / which group of RX outputs this thread takes (one-wave form: rows side by side)
    const int wr = NT_ == 256 ? tid >> 6 : tid / C::NG, g = NT_ == 256 ? tid & 63 : tid % C::NG;
    for (int base = 0; base < C::EH; base += C::UNROLL) {
#pragma unroll
        for (int j = 0; j < C::UNROLL; j++) {
            const int i = base + j;                   // eigenvalue row this covariance row completes
            const int K = (BS - 1 + j) % BS;          // ring slot: holds the product blockSize rows up
            float p[3];
            cov_row(BS - 1 + i, p[0], p[1], p[2]);
#pragma unroll
            for (int q = 0; q < 3; q++) {
                V[q] = (V[q] + (double)p[q]) - (double)ring[q][K];
                ring[q][K] = p[q];
                Vb[((j % C::R) * 3 + q) * C::VP + tid] = V[q];
            }
            __builtin_amdgcn_sched_barrier(0);           // rows stay in order: the loads of 20 unrolled rows are not hoisted
            if (j % C::R != C::R - 1) continue;
            const int eb = i / C::R;                  // batch: eigenvalue rows eb*R .. eb*R + R-1
            __syncthreads();
            if (g < C::NG && wr < C::R) {
                double Sm[3][C::RX];
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    double v[C::RX + BS - 1];
                    const double* src = Vb + (wr * 3 + q) * C::VP + C::RX * g;
#pragma unroll
                    for (int k = 0; k < C::RX + BS - 1; k++) v[k] = src[k];
                    window_sums<C::RX, BS>(v, Sm[q]);
                    __builtin_amdgcn_sched_barrier(0);   // one plane's 13 doubles at a time, not all three
                }
                const int er = eb * C::R + wr;
                float e[C::RX];
#pragma unroll
                for (int k = 0; k < C::RX; k++) e[k] = min_eig_of(Sm[0][k], Sm[1][k], Sm[2][k]);
                *reinterpret_cast<float4*>(Er + (er & 7) * C::NT + C::RX * g) = make_float4(e[0], e[1], e[2], e[3]);
                const int y = y0 - 1 + er;
                if (er >= 1 && er <= C::SH && y < h) {
#pragma unroll
                    for (int k = 0; k < C::RX; k++) {
                        const int ce = C::RX * g + k, x = x0 - 1 + ce;
                        if (ce >= 1 && ce <= C::TW && x < w) {
                            if (eig_out) eig_out[(size_t)y * w + x] = e[k];
                            if (!mask || mask[(size_t)y * mask_pitch + x]) {
                                const unsigned key = ordered_key(e[k]);
                                best = key > best ? key : best;
                            }
                        }
                    }
                }
            }
            __syncthreads();
            {
                // 3x3 non-max test of eigenvalue rows eb*R - 1 .. eb*R + 2, column tid
                const int x = x0 - 1 + tid;
                if (tid >= 1 && tid <= C::TW && x >= 1 && x < w - 1) {
                    float m3[6], mid[6], side[6];
#pragma unroll
                    for (int k = 0; k < 6; k++) {
                        const float* row = Er + ((eb * C::R - 2 + k) & 7) * C::NT + tid;
                        const float l = row[-1], c = row[0], r = row[1];
                        mid[k] = c;
                        side[k] = fmaxf(l, r);
                        m3[k] = fmaxf(side[k], c);
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int c = eb * C::R - 1 + q, y = y0 - 1 + c;
                        if (c < 1 || c > C::SH || y < 1 || y >= h - 1) continue;
                        const float v = mid[q + 1];
                        if (!(v > 0.f)) continue;
                        const float m = fmaxf(fmaxf(m3[q], m3[q + 2]), side[q + 1]);
                        if (v < m) continue;
                        if (mask && !mask[(size_t)y * mask_pitch + x]) continue;
                        const int sub = (c - 1) / C::SUB;
                        const int pos = atomicAdd(&s_cnt[sub], 1);
                        raw[((size_t)bid * C::NSUB + sub) * (C::TW * C::SUB) + pos] =
                            ((unsigned long long)ordered_key(v) << 32) | pack_xy(x, y);
                    }
                }
            }
        }
    }
    publish_max(max_key, best, tid);
    __syncthreads();
    if (tid < C::NSUB) blk_count[bid * C::NSUB + tid] = s_cnt[tid];
}

template <int BS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(80))) void k_eig_strip1(const uint8_t* __restrict__ img, int w, int h, int pitch, float k0,
                                                    float k1, const uint8_t* __restrict__ mask, int mask_pitch,
                                                    unsigned* __restrict__ max_key,
                                                    unsigned long long* __restrict__ raw, int* __restrict__ blk_count,
                                                    float* __restrict__ eig_out)
{
    using C = StripCfg<BS, 64>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ int s_cnt[C::NSUB];
    const int ys = (int)blockIdx.y * C::SH - 1 - C::AN;
    if (ys - 1 >= 0 && ys + C::NROWS <= h - 1)
        strip_body<BS, false, 64>(img, w, h, pitch, k0, k1, mask, mask_pitch, max_key, raw, blk_count, eig_out, smem, s_cnt);
    else
        strip_body<BS, true, 64>(img, w, h, pitch, k0, k1, mask, mask_pitch, max_key, raw, blk_count, eig_out, smem, s_cnt);
}

// Registers: what the kernel needs at blockSize 10 is 144 VGPRs.  Until late in round 4 it was held to 128 for a fourth wave
// per SIMD -- 18 spilled VGPRs, 76 B of scratch per lane: 19 MB of scratch written per launch (WRITE_SIZE) -- which is
// faster ALONE (81 against 98 us) and slower where it counts: beside a tracker launch the tracker decides how many of its
// waves fit, and without scratch C2 runs at 6 900-6 940 pairs/s instead of 6 620-6 650 (same-box A/B,
// profiles/r04_ab_corner_kernel_registers.txt; REF equal, C5 +1 %).  blockSize 3 / 5 / 7 need 87 / 110 / 124 and keep four waves.
template <int BS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_eig_strip(const uint8_t* __restrict__ img, int w, int h, int pitch, float k0,
                                                    float k1, const uint8_t* __restrict__ mask, int mask_pitch,
                                                    unsigned* __restrict__ max_key,
                                                    unsigned long long* __restrict__ raw, int* __restrict__ blk_count,
                                                    float* __restrict__ eig_out)
{
    using C = StripCfg<BS>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ int s_cnt[C::NSUB];
    const int ys = (int)blockIdx.y * C::SH - 1 - C::AN;
    // pixel rows ys - 1 .. ys + NROWS are read by the rolling form
    if (ys - 1 >= 0 && ys + C::NROWS <= h - 1)
        strip_body<BS, false>(img, w, h, pitch, k0, k1, mask, mask_pitch, max_key, raw, blk_count, eig_out, smem, s_cnt);
    else
        strip_body<BS, true>(img, w, h, pitch, k0, k1, mask, mask_pitch, max_key, raw, blk_count, eig_out, smem, s_cnt);
}

// workgroup-aggregated append: every thread offers at most one key per call; one global atomic per call
__device__ __forceinline__ void block_append(bool keep, unsigned long long key, unsigned long long* out,
                                             int* out_count, int* s_cnt, int* s_base)
{
    if (threadIdx.x == 0) *s_cnt = 0;
    __syncthreads();
    const int pos = keep ? atomicAdd(s_cnt, 1) : 0;
    __syncthreads();
    if (threadIdx.x == 0 && *s_cnt) *s_base = atomicAdd(out_count, *s_cnt);
    __syncthreads();
    if (keep) out[*s_base + pos] = key;
    __syncthreads();
}

// regions -> flat list of the candidates above max * qualityLevel (minDistance < 1 path: everything is sorted)
__global__ __launch_bounds__(256) void k_flatten(CandSrc src, const unsigned* __restrict__ max_key, double quality,
                                                 unsigned long long* __restrict__ cand, int* __restrict__ cand_count)
{
    __shared__ int s_cnt, s_base;
    const float thr = threshold_of(max_key, quality);
    for (int b = blockIdx.x; b < src.nblk; b += gridDim.x) {
        const int cnt = src.blk_count[b];
        for (int i0 = 0; i0 < cnt; i0 += 256) {
            const int i = i0 + threadIdx.x;
            unsigned long long key = 0;
            bool keep = false;
            if (i < cnt) {
                key = src.keys[(size_t)b * src.region + i];
                keep = key_to_float((unsigned)(key >> 32)) > thr;
            }
            block_append(keep, key, cand, cand_count, &s_cnt, &s_base);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K6 generic: any blockSize, writes the eigenvalue map.  LDS: cov[3][eh][ew] f32, hs[3][eh][TW] f64.
// ------------------------------------------------------------------------------------------------
constexpr int EIG_TW = 64;
constexpr int EIG_TH = 16;

__global__ __launch_bounds__(256) void k_min_eig(const uint8_t* __restrict__ img, int w, int h, int pitch, int bs,
                                                 float k0, float k1, float* __restrict__ eig,
                                                 const uint8_t* __restrict__ mask, int mask_pitch,
                                                 unsigned* __restrict__ max_key, int variant)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int anchor = bs / 2;
    const int ew = EIG_TW + bs - 1, eh = EIG_TH + bs - 1;
    double* hs = reinterpret_cast<double*>(smem);
    float* cov = reinterpret_cast<float*>(smem + sizeof(double) * 3 * eh * EIG_TW);
    const int x0 = blockIdx.x * EIG_TW, y0 = blockIdx.y * EIG_TH;
    const int tid = threadIdx.x;
    for (int i = tid; i < ew * eh; i += 256) {
        const int ey = i / ew, ex = i - ey * ew;
        const int rx = reflect101(x0 - anchor + ex, w), ry = reflect101(y0 - anchor + ey, h);
        const int xm = reflect101(rx - 1, w), xp = reflect101(rx + 1, w);
        const int ym = reflect101(ry - 1, h), yp = reflect101(ry + 1, h);
        const uint8_t* r0 = img + (size_t)ym * pitch;
        const uint8_t* r1 = img + (size_t)ry * pitch;
        const uint8_t* r2 = img + (size_t)yp * pitch;
        float xx, xy, yy;
        sobel_cov((float)r0[xm], (float)r0[rx], (float)r0[xp], (float)r1[xm], (float)r1[xp], (float)r2[xm],
                  (float)r2[rx], (float)r2[xp], k0, k1, xx, xy, yy, variant);
        cov[i] = xx;
        cov[ew * eh + i] = xy;
        cov[2 * ew * eh + i] = yy;
    }
    __syncthreads();
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
    unsigned best = 0;
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
        const float v = min_eig_of(s0, s1, s2, variant);
        eig[(size_t)y * w + x] = v;
        if (!mask || mask[(size_t)y * mask_pitch + x]) {
            const unsigned k = ordered_key(v);
            best = k > best ? k : best;
        }
    }
    publish_max(max_key, best, tid);
}

// K7 generic: threshold + 3x3 dilate equality on the materialised map; a workgroup covers a 256 x 8 pixel band
// and owns region (blockIdx) of the candidate buffer.
constexpr int NMS_ROWS = 8;
__global__ __launch_bounds__(256) void k_nms_collect(const float* __restrict__ eig, int w, int h,
                                                     const uint8_t* __restrict__ mask, int mask_pitch,
                                                     const unsigned* __restrict__ max_key, double quality,
                                                     unsigned long long* __restrict__ raw, int* __restrict__ blk_count)
{
    __shared__ int list_n;
    const int tid = threadIdx.x;
    if (tid == 0) list_n = 0;
    __syncthreads();
    const int bid = blockIdx.y * gridDim.x + blockIdx.x;
    unsigned long long* region = raw + (size_t)bid * (256 * NMS_ROWS);
    const float thr = threshold_of(max_key, quality);
    const int x = blockIdx.x * 256 + tid + 1;
    for (int r = 0; r < NMS_ROWS; r++) {
        const int y = blockIdx.y * NMS_ROWS + r + 1;
        if (x >= w - 1 || y >= h - 1) continue;
        const float v = eig[(size_t)y * w + x];
        if (!(v > thr) || v == 0.f) continue;
        if (mask && !mask[(size_t)y * mask_pitch + x]) continue;
        float m = 0.f;  // dilate of the TOZERO-thresholded map (includes the centre)
#pragma unroll
        for (int dy = -1; dy <= 1; dy++)
#pragma unroll
            for (int dx = -1; dx <= 1; dx++) {
                float q = eig[(size_t)(y + dy) * w + (x + dx)];
                q = q > thr ? q : 0.f;
                m = q > m ? q : m;
            }
        if (v != m) continue;
        region[atomicAdd(&list_n, 1)] = ((unsigned long long)ordered_key(v) << 32) | pack_xy(x, y);
    }
    __syncthreads();
    if (tid == 0) blk_count[bid] = list_n;
}

// ------------------------------------------------------------------------------------------------
// K8 helpers.
// Every kernel of the min-distance stage is launched as ONE-WAVE workgroups (CT = 64 threads) with at most 8 KB of
// LDS and no more than 128 VGPRs: that is the footprint of a single k_lk_fast wave, so while a tracker launch
// owns the chip (it holds every VGPR) such a workgroup fits into the hole any retiring tracker wave leaves and
// the stage progresses beside it; wider workgroups wait for the tracker's tail (tools/ubench/corun.hip:
// 64 threads / 4 KB LDS 30-60 us beside a register-saturating kernel, >= 128 threads or 16 KB 250-340 us).
// ------------------------------------------------------------------------------------------------
constexpr int CT = 64;
constexpr int SCAN_CHUNK = 2048;   // cells per workgroup of k_scan; k_cell_count also keeps per-chunk totals
// The chunk totals are hot atomic targets: one per 128-byte line (atomics on one line serialise in its L2 channel,
// ~90 per us), and a wave adds its contributions to a chunk with one atomic.
constexpr int CHUNK_TOT_STRIDE = 32;

__device__ __forceinline__ void chunk_add(int* __restrict__ chunk_tot, bool active, int chunk)
{
    unsigned long long pending = __ballot(active);
    while (pending) {
        const int leader = __ffsll((long long)pending) - 1;
        const int c = __shfl(chunk, leader);
        const unsigned long long same = __ballot(active && chunk == c) & pending;
        if ((int)threadIdx.x == leader) atomicAdd(&chunk_tot[c * CHUNK_TOT_STRIDE], __popcll(same));
        pending &= ~same;
    }
}

__global__ __launch_bounds__(CT) void k_cell_count(CandSrc src, const unsigned* __restrict__ max_key, double quality,
                                                   const unsigned* __restrict__ prune_key, int w, int cell, int gw,
                                                   int* __restrict__ cell_count, int* __restrict__ chunk_tot)
{
    const float thr = threshold_of(max_key, quality);
    const unsigned pk = *prune_key;
    for (int b = blockIdx.x; b < src.nblk; b += gridDim.x) {
        const int cnt = src.blk_count[b];
        for (int i0 = 0; i0 < cnt; i0 += CT) {
            const int i = i0 + threadIdx.x;
            bool keep = false;
            int c = 0;
            if (i < cnt) {
                const unsigned long long key = src.keys[(size_t)b * src.region + i];
                keep = (unsigned)(key >> 32) >= pk && key_to_float((unsigned)(key >> 32)) > thr;
                const unsigned idx = (unsigned)key;
                const int y = (int)(idx >> 16), x = (int)(idx & 0xffffu);
                c = (y / cell) * gw + (x / cell);
            }
            if (keep) atomicAdd(&cell_count[c], 1);
            chunk_add(chunk_tot, keep, c / SCAN_CHUNK);
        }
    }
}

// ---- top-K pruning when maxCorners caps the output ------------------------------------------------
// Whether a candidate is accepted depends only on STRONGER candidates, so the accepted candidates among the K
// strongest are exactly the greedy-accepted ones among them; if they number >= maxCorners, the weaker
// candidates can never appear in the output and need not be binned, relaxed or sorted.  K is found from a
// 16384-bin histogram of the response key (its top 14 bits); the host checks "accepted >= maxCorners" at its
// one synchronisation point and reruns unpruned otherwise.
constexpr int KEY_SHIFT = 18;                 // bin width: sign, exponent and 5 mantissa bits of the key
constexpr int KEY_BINS = 2048;                // bins counted DOWN from the bin of the maximum: 64 octaves
// Responses cluster in a few hundred bins, so a global-memory histogram would serialise on a handful of L2
// atomic addresses: every workgroup histograms into LDS (8 KB) and flushes its non-empty bins.
__device__ __forceinline__ int key_bin(unsigned key, unsigned max_key)
{
    const int b = (int)(max_key >> KEY_SHIFT) - (int)(key >> KEY_SHIFT);
    return b < 0 ? 0 : (b >= KEY_BINS ? KEY_BINS - 1 : b);
}

__global__ __launch_bounds__(CT) void k_key_hist(CandSrc src, const unsigned* __restrict__ max_key, double quality,
                                                 unsigned* __restrict__ hist)
{
    __shared__ unsigned lh[KEY_BINS];
    for (int i = threadIdx.x; i < KEY_BINS; i += CT) lh[i] = 0;
    __syncthreads();
    const float thr = threshold_of(max_key, quality);
    const unsigned mk = *max_key;
    for (int b = blockIdx.x; b < src.nblk; b += gridDim.x) {
        const int cnt = src.blk_count[b];
        for (int i = threadIdx.x; i < cnt; i += CT) {
            const unsigned k = (unsigned)(src.keys[(size_t)b * src.region + i] >> 32);
            if (key_to_float(k) > thr) atomicAdd(&lh[key_bin(k, mk)], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < KEY_BINS; i += CT) {
        const unsigned v = lh[i];
        if (v) atomicAdd(&hist[i], v);
    }
}

// prune_key = lowest key of the strongest bins that together hold >= want candidates (0 = keep everything).
// One wave; lane l owns bins [32 l, 32 l + 32), bin 0 being the strongest.
__global__ __launch_bounds__(CT) void k_key_select(const unsigned* __restrict__ hist, unsigned want,
                                                   const unsigned* __restrict__ max_key,
                                                   unsigned* __restrict__ prune_key)
{
    constexpr int PER = KEY_BINS / CT;
    const int lane = threadIdx.x;
    unsigned s = 0;
    for (int i = 0; i < PER; i++) s += hist[lane * PER + i];
    unsigned incl = s;   // candidates in this lane's bins and all stronger ones
#pragma unroll
    for (int o = 1; o < CT; o <<= 1) {
        const unsigned v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    const unsigned total = __shfl(incl, CT - 1);
    if (lane == 0 && total < want) *prune_key = 0u;
    const unsigned above = incl - s;
    if (above < want && incl >= want) {
        unsigned run = above;
        for (int i = 0; i < PER; i++) {
            run += hist[lane * PER + i];
            if (run >= want) {
                const int bin = lane * PER + i;   // keep bins 0..bin
                const int top = (int)(*max_key >> KEY_SHIFT);
                // the last bin also collects everything weaker: keeping it means keeping all
                *prune_key = (bin >= KEY_BINS - 1 || top - bin <= 0) ? 0u : (unsigned)(top - bin) << KEY_SHIFT;
                break;
            }
        }
    }
}

// exclusive scan of the cell counts, SCAN_CHUNK cells per one-wave workgroup: the offset of a chunk is the sum of the
// chunk totals before it (kept by k_cell_count), inside the chunk 8 rounds of 4 cells per lane.  start[n] = total.
__global__ __launch_bounds__(CT) void k_scan(const int* __restrict__ count, const int* __restrict__ chunk_tot,
                                             int* __restrict__ start, int n)
{
    const int lane = threadIdx.x;
    const int lo = blockIdx.x * SCAN_CHUNK;
    int run = 0;
    for (int c = lane; c < (int)blockIdx.x; c += CT) run += chunk_tot[c * CHUNK_TOT_STRIDE];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) run += __shfl_xor(run, o);
    for (int r = 0; r < SCAN_CHUNK / (4 * CT); r++) {
        const int i0 = lo + r * 4 * CT + 4 * lane;
        int c[4], s = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            c[k] = i0 + k < n ? count[i0 + k] : 0;
            s += c[k];
        }
        int inc = s;
#pragma unroll
        for (int o = 1; o < CT; o <<= 1) {
            const int v = __shfl_up(inc, o);
            if (lane >= o) inc += v;
        }
        int at = run + inc - s;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (i0 + k < n) start[i0 + k] = at;
            at += c[k];
            if (i0 + k == n - 1) start[n] = at;
        }
        run += __shfl(inc, CT - 1);
    }
    if (n == 0 && blockIdx.x == 0 && lane == 0) start[0] = 0;
}

__global__ __launch_bounds__(CT) void k_cell_fill(CandSrc src, const unsigned* __restrict__ max_key, double quality,
                                                   const unsigned* __restrict__ prune_key, int w, int cell, int gw,
                                                   const int* __restrict__ cell_start, int* __restrict__ cell_fill,
                                                   unsigned long long* __restrict__ cell_cand,
                                                   uint8_t* __restrict__ state)
{
    const float thr = threshold_of(max_key, quality);
    const unsigned pk = *prune_key;
    for (int b = blockIdx.x; b < src.nblk; b += gridDim.x) {
        const int cnt = src.blk_count[b];
        for (int i = threadIdx.x; i < cnt; i += CT) {
            const unsigned long long key = src.keys[(size_t)b * src.region + i];
            if ((unsigned)(key >> 32) < pk || !(key_to_float((unsigned)(key >> 32)) > thr)) continue;
            const unsigned idx = (unsigned)key;
            const int y = (int)(idx >> 16), x = (int)(idx & 0xffffu);
            const int c = (y / cell) * gw + (x / cell);
            const int pos = cell_start[c] + atomicAdd(&cell_fill[c], 1);
            cell_cand[pos] = key;
            state[pos] = 0;
        }
    }
}

// Relaxation.  A thread owns one candidate.  Its first look scans the 3x3 cells and keeps the indices of the
// stronger candidates within minDistance that are still undecided (its "blockers") in LDS; after that it only
// polls those.  States are read and written with L1-bypassing (system-scope relaxed) accesses and only ever
// move 0 -> 1 or 0 -> 2 on final facts, so progress made by other workgroups is seen as it happens and a stale
// read merely costs another poll.  Spins are bounded; launch_counters[r] = candidates still undecided after
// launch r, and launch r+1 picks them up (it returns at once when that count is zero).
constexpr int SUP_K = 24;
constexpr int SUP_SPINS = 48;
__global__ __launch_bounds__(CT) void k_suppress(const unsigned long long* __restrict__ cell_cand,
                                                  const int* __restrict__ n_ptr, int cell, int gw, int gh,
                                                  const int* __restrict__ cell_start, uint8_t* state, double md2,
                                                  int* __restrict__ launch_counters, int r)
{
    __shared__ int blockers[SUP_K * CT];
    if (r > 0 && launch_counters[r - 1] == 0) return;
    const int n = *n_ptr;
    int* mine = blockers + threadIdx.x;   // entry q at mine[q * CT]: conflict-free
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (__atomic_load_n(&state[i], __ATOMIC_RELAXED)) continue;
        const unsigned long long key = cell_cand[i];
        const unsigned idx = (unsigned)key;
        const int y = (int)(idx >> 16), x = (int)(idx & 0xffffu);
        const int xc = x / cell, yc = y / cell;
        const int x1 = max(0, xc - 1), y1 = max(0, yc - 1), x2 = min(gw - 1, xc + 1), y2 = min(gh - 1, yc + 1);
        int cnt = 0;
        bool rejected = false, overflow = false;
        for (int yy = y1; yy <= y2 && !rejected; yy++) {
            const int b = cell_start[yy * gw + x1], e = cell_start[yy * gw + x2 + 1];  // the 3 cells are contiguous
            for (int j = b; j < e; j++) {
                const unsigned long long kj = cell_cand[j];
                if (kj <= key) continue;  // only stronger candidates matter (keys are unique)
                const unsigned ij = (unsigned)kj;
                const int yj = (int)(ij >> 16), xj = (int)(ij & 0xffffu);
                const float dx = (float)(x - xj), dy = (float)(y - yj);
                if (!((double)(dx * dx + dy * dy) < md2)) continue;
                const uint8_t sj = __atomic_load_n(&state[j], __ATOMIC_RELAXED);
                if (sj == 1) { rejected = true; break; }
                if (sj == 0) {
                    if (cnt < SUP_K) mine[CT * cnt++] = j;
                    else overflow = true;
                }
            }
        }
        bool decided = false;
        if (rejected) { __atomic_store_n(&state[i], (uint8_t)2, __ATOMIC_RELAXED); decided = true; }
        else if (cnt == 0 && !overflow) { __atomic_store_n(&state[i], (uint8_t)1, __ATOMIC_RELAXED); decided = true; }
        for (int spin = 0; spin < SUP_SPINS && !decided && !overflow; spin++) {
            __builtin_amdgcn_s_sleep(8);
            int k = 0;
            for (int q = 0; q < cnt; q++) {
                const int j = mine[CT * q];
                const uint8_t sj = __atomic_load_n(&state[j], __ATOMIC_RELAXED);
                if (sj == 1) { rejected = true; break; }
                if (sj == 0) mine[CT * k++] = j;
            }
            cnt = k;
            if (rejected) { __atomic_store_n(&state[i], (uint8_t)2, __ATOMIC_RELAXED); decided = true; }
            else if (cnt == 0) { __atomic_store_n(&state[i], (uint8_t)1, __ATOMIC_RELAXED); decided = true; }
        }
        if (!decided) atomicAdd(&launch_counters[r], 1);
    }
}

// zero every counter a detection uses, in one launch (each hipMemsetAsync is a ~5 us kernel of its own)
__global__ void k_detect_reset(int* __restrict__ cell_count, int* __restrict__ cell_fill, int ncell,
                               int* __restrict__ chunk_tot, int* __restrict__ undecided, int* __restrict__ acc_count,
                               int* __restrict__ cand_count, unsigned* __restrict__ max_key,
                               unsigned* __restrict__ key_hist, unsigned* __restrict__ prune_key, int full)
{
    const int i0 = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = i0; i <= ncell; i += gridDim.x * blockDim.x) {
        cell_count[i] = 0;
        if (i < ncell) cell_fill[i] = 0;
        if (i % SCAN_CHUNK == 0) chunk_tot[(i / SCAN_CHUNK) * CHUNK_TOT_STRIDE] = 0;
    }
    if (full & 1)
        for (int i = i0; i < KEY_BINS; i += gridDim.x * blockDim.x) key_hist[i] = 0;
    if (i0 < 8) undecided[i0] = 0;
    if (i0 == 0) {
        *acc_count = 0;
        *cand_count = 0;
        *prune_key = 0;
        if (full & 2) *max_key = 0;
    }
}

__global__ __launch_bounds__(CT) void k_gather_accepted(const unsigned long long* __restrict__ cell_cand,
                                                         const int* __restrict__ n_ptr,
                                                         const uint8_t* __restrict__ state,
                                                         unsigned long long* __restrict__ acc,
                                                         int* __restrict__ acc_count)
{
    __shared__ int s_cnt, s_base;
    const int n = *n_ptr;
    for (int i0 = blockIdx.x * CT; i0 < n; i0 += gridDim.x * CT) {
        const int i = i0 + threadIdx.x;
        const bool keep = i < n && state[i] == 1;
        block_append(keep, keep ? cell_cand[i] : 0ull, acc, acc_count, &s_cnt, &s_base);
    }
}

__global__ void k_emit(const unsigned long long* __restrict__ keys, int n, int w, float* __restrict__ xy)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned idx = (unsigned)keys[i];
    const int y = (int)(idx >> 16), x = (int)(idx & 0xffffu);
    xy[2 * i] = (float)x;
    xy[2 * i + 1] = (float)y;
}

// The tail of a detection whose corners start a segment, in ONE launch instead of three (k_emit, k_detect_reset,
// k_seg_init): every launch of the tail has to find room beside a tracker launch -- or beside a corner kernel that has the
// chip to itself -- and the next tracker launch waits for the last of them.  Blocks [0, emit_blocks): corner i from its
// sorted key -> the corner list AND the new segment's tables (position, alive flag, vertex 0 of its track); the blocks
// behind them: the counters of the detector set for its next detection (mode as launch_detect_reset).
__global__ __launch_bounds__(CT) void k_tail(const unsigned long long* __restrict__ keys, int n, float* __restrict__ corners,
                                             float* __restrict__ seg_xy, uint8_t* __restrict__ seg_alive,
                                             float* __restrict__ seg_tracks, int max_vert, int emit_blocks,
                                             int* __restrict__ cell_count, int* __restrict__ cell_fill, int ncell,
                                             int* __restrict__ chunk_tot, int* __restrict__ undecided,
                                             int* __restrict__ acc_count, int* __restrict__ cand_count,
                                             unsigned* __restrict__ max_key, unsigned* __restrict__ key_hist,
                                             unsigned* __restrict__ prune_key, int full)
{
    if ((int)blockIdx.x < emit_blocks) {
        const int i = blockIdx.x * CT + threadIdx.x;
        if (i >= n) return;
        const unsigned idx = (unsigned)keys[i];
        const float x = (float)(int)(idx & 0xffffu), y = (float)(int)(idx >> 16);
        corners[2 * i] = x; corners[2 * i + 1] = y;
        seg_xy[2 * i] = x; seg_xy[2 * i + 1] = y;
        seg_alive[i] = 1;
        seg_tracks[((size_t)i * max_vert) * 2] = x;
        seg_tracks[((size_t)i * max_vert) * 2 + 1] = y;
        return;
    }
    const int i0 = ((int)blockIdx.x - emit_blocks) * CT + threadIdx.x, stride = ((int)gridDim.x - emit_blocks) * CT;
    for (int i = i0; i <= ncell; i += stride) {
        cell_count[i] = 0;
        if (i < ncell) cell_fill[i] = 0;
        if (i % SCAN_CHUNK == 0) chunk_tot[(i / SCAN_CHUNK) * CHUNK_TOT_STRIDE] = 0;
    }
    if (full & 1)
        for (int i = i0; i < KEY_BINS; i += stride) key_hist[i] = 0;
    if (i0 < 8) undecided[i0] = 0;
    if (i0 == 0) {
        *acc_count = 0;
        *cand_count = 0;
        *prune_key = 0;
        if (full & 2) *max_key = 0;
    }
}

void sobel_scale(int block_size, float* k0, float* k1)
{
    double scale = (double)(1 << 2) * block_size;
    scale *= 255.0;
    scale = 1.0 / scale;
    *k1 = (float)(1.0 * scale);
    *k0 = (float)(2.0 * scale);
}

template <int BS>
void launch_fused(hipStream_t s, const Level& img, float k0, float k1, const uint8_t* mask, int mask_pitch,
                  unsigned* max_key, unsigned long long* raw, int* blk_count, float* eig_out, CandSrc* src)
{
    using C = EigCfg<BS>;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_eig_nms<BS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            C::LDS_BYTES);
        attr_set = true;
    }
    dim3 grid((img.w + C::TW - 1) / C::TW, (img.h + C::TH - 1) / C::TH);
    hipLaunchKernelGGL((k_eig_nms<BS>), grid, dim3(256), C::LDS_BYTES, s, img.ptr, img.w, img.h, img.pitch, k0, k1, mask,
                       mask_pitch, max_key, raw, blk_count, eig_out);
    src->keys = raw;
    src->blk_count = blk_count;
    src->nblk = (int)(grid.x * grid.y);
    src->region = C::TW * C::TH;
}

template <int BS>
void launch_strip(hipStream_t s, const Level& img, float k0, float k1, const uint8_t* mask, int mask_pitch,
                  unsigned* max_key, unsigned long long* raw, int* blk_count, float* eig_out, CandSrc* src, bool one_wave)
{
    if (one_wave) {
        using C = StripCfg<BS, 64>;
        dim3 grid((img.w + C::TW - 1) / C::TW, (img.h + C::SH - 1) / C::SH);
        hipLaunchKernelGGL((k_eig_strip1<BS>), grid, dim3(C::NT), C::LDS_BYTES, s, img.ptr, img.w, img.h, img.pitch, k0, k1, mask,
                           mask_pitch, max_key, raw, blk_count, eig_out);
        src->nblk = (int)(grid.x * grid.y) * C::NSUB;
        src->region = C::TW * C::SUB;
    } else {
        using C = StripCfg<BS>;
        dim3 grid((img.w + C::TW - 1) / C::TW, (img.h + C::SH - 1) / C::SH);
        hipLaunchKernelGGL((k_eig_strip<BS>), grid, dim3(C::NT), C::LDS_BYTES, s, img.ptr, img.w, img.h, img.pitch, k0, k1, mask,
                           mask_pitch, max_key, raw, blk_count, eig_out);
        src->nblk = (int)(grid.x * grid.y) * C::NSUB;
        src->region = C::TW * C::SUB;
    }
    src->keys = raw;
    src->blk_count = blk_count;
}

// regions / keys the strip layout needs, whichever blockSize is asked for later
template <int BS, int NT_>
void strip_geometry1(int w, int h, size_t* regions, size_t* keys)
{
    using C = StripCfg<BS, NT_>;
    const size_t r = (size_t)((w + C::TW - 1) / C::TW) * ((h + C::SH - 1) / C::SH) * C::NSUB;
    const size_t k = r * (size_t)(C::TW * C::SUB);
    *regions = r > *regions ? r : *regions;
    *keys = k > *keys ? k : *keys;
}
template <int BS>
void strip_geometry(int w, int h, size_t* regions, size_t* keys)
{
    strip_geometry1<BS, 256>(w, h, regions, keys);
    strip_geometry1<BS, 64>(w, h, regions, keys);
}
size_t strip_regions(int w, int h)
{
    size_t r = 0, k = 0;
    strip_geometry<3>(w, h, &r, &k); strip_geometry<5>(w, h, &r, &k); strip_geometry<7>(w, h, &r, &k); strip_geometry<10>(w, h, &r, &k);
    return r;
}
size_t strip_keys(int w, int h)
{
    size_t r = 0, k = 0;
    strip_geometry<3>(w, h, &r, &k); strip_geometry<5>(w, h, &r, &k); strip_geometry<7>(w, h, &r, &k); strip_geometry<10>(w, h, &r, &k);
    return k;
}

}  // namespace

size_t min_eig_lds_bytes(int block_size)
{
    const int ew = EIG_TW + block_size - 1, eh = EIG_TH + block_size - 1;
    return sizeof(double) * 3 * eh * EIG_TW + sizeof(float) * 3 * eh * ew;
}

bool fused_block_size(int bs) { return bs == 3 || bs == 5 || bs == 7 || bs == 10; }

// capacity (in keys) the region layout needs for a w x h frame, whichever kernel produces the candidates
size_t candidate_capacity(int w, int h)
{
    const size_t fused = (size_t)((w + 63) / 64) * ((h + 15) / 16) * (64 * 16);
    const size_t generic = (size_t)((w + 255) / 256) * ((h + NMS_ROWS - 1) / NMS_ROWS) * (256 * NMS_ROWS);
    const size_t fast = fast_key_capacity(w, h), strip = strip_keys(w, h);
    size_t m = fused > generic ? fused : generic;
    m = m > fast ? m : fast;
    return (m > strip ? m : strip) + 1024;
}
size_t candidate_blocks(int w, int h)
{
    const size_t a = (size_t)((w + 63) / 64) * ((h + 15) / 16), b = fast_regions(w, h), c = strip_regions(w, h);
    const size_t m = a > b ? a : b;
    return (m > c ? m : c) + 16;
}

// K6 alone, writing the map with the any-blockSize kernel.
void launch_min_eig(hipStream_t s, const Level& img, int block_size, float* eig, const uint8_t* mask,
                    int mask_pitch, unsigned* max_key, int variant)
{
    float k0, k1;
    sobel_scale(block_size, &k0, &k1);
    const size_t lds = min_eig_lds_bytes(block_size);
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_min_eig), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
        attr_lds = lds;
    }
    dim3 grid((img.w + EIG_TW - 1) / EIG_TW, (img.h + EIG_TH - 1) / EIG_TH);
    hipLaunchKernelGGL(k_min_eig, grid, dim3(256), lds, s, img.ptr, img.w, img.h, img.pitch, block_size, k0, k1, eig,
                       mask, mask_pitch, max_key, variant);
}

static CandSrc src_of(const DetectScratch& D)
{
    CandSrc c;
    c.keys = D.raw;
    c.blk_count = D.blk_count;
    c.nblk = D.src_nblk;
    c.region = D.src_region;
    return c;
}

// First launch of every detection: zero the counters (ncell = 0 when minDistance < 1).
// full = also the response maximum and the key histogram (i.e. everything a NEW detection needs); !full = only
// what a re-run of the min-distance stage on the same candidates needs.
void launch_detect_reset(hipStream_t s, DetectScratch& D, int ncell, int mode)
{
    int blocks = (ncell + CT) / CT;
    if ((mode & 1) && blocks < KEY_BINS / CT) blocks = KEY_BINS / CT;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_detect_reset, dim3(blocks), dim3(CT), 0, s, D.cell_count, D.cell_fill, ncell, D.chunk_tot, D.undecided,
                       D.acc_count, D.cand_count, D.max_key, D.key_hist, D.prune_key, mode);
}

// Candidate collection (K6+K7) into regions of D.raw (stream order, no host sync).
void launch_candidates(hipStream_t s, DetectScratch& D, const Level& img, int block_size, const uint8_t* mask,
                       int mask_pitch, double quality, bool use_generic, float* eig_out_or_null, int variant, bool beside_tracker)
{
    if (variant) use_generic = true;     // the named variants live in the any-blockSize kernel only
    unsigned long long* raw = D.raw;
    CandSrc g_src{};
    // ICELK_TWO_PASS_CORNERS=1: the two-pass form (k_corners_fast.hip: integer bracket of the map + exact arithmetic at the
    // possible maxima) -- a third statement of the arithmetic, bit-identical, measured and not the default
    const bool two_pass = D.acand != nullptr && getenv("ICELK_TWO_PASS_CORNERS") != nullptr;   // scratch: at icelk_create, under the same switch
    if (two_pass && !use_generic && fused_block_size(block_size) && !eig_out_or_null &&
        launch_candidates_fast(s, D, img, block_size, mask, mask_pitch, quality))
        return;
    // ICELK_TILE_CORNERS=1: round 2's 64x16-tile kernel instead of the strip kernel (same lists; kept for comparison)
    const bool tiles = getenv("ICELK_TILE_CORNERS") != nullptr;
    if (!use_generic && fused_block_size(block_size) && !tiles) {
        float k0, k1;
        sobel_scale(block_size, &k0, &k1);
        // ICELK_STRIP_WAVES=1: one-wave strips (k_eig_strip1) for the corner kernel that runs ahead of its detection, beside a
        // tracker launch (icelk_seg_detect_prepare); =11: everywhere.  Default: round 3's four-wave strips -- measured on one
        // box, C2 / REF / C5: 5 950 / 766 / 567 pairs/s with one-wave strips against 6 000 / 772 / 569 (the one-wave
        // workgroups do get onto the CUs at once, and the tracker launch beside them takes 292 us instead of 273)
        static const char* sw = getenv("ICELK_STRIP_WAVES");
        const int swv = sw ? atoi(sw) : 0;
        const bool one_wave = swv == 11 || (swv == 1 && beside_tracker);
        switch (block_size) {
            case 3: launch_strip<3>(s, img, k0, k1, mask, mask_pitch, D.max_key, raw, D.blk_count, eig_out_or_null, &g_src, one_wave); break;
            case 5: launch_strip<5>(s, img, k0, k1, mask, mask_pitch, D.max_key, raw, D.blk_count, eig_out_or_null, &g_src, one_wave); break;
            case 7: launch_strip<7>(s, img, k0, k1, mask, mask_pitch, D.max_key, raw, D.blk_count, eig_out_or_null, &g_src, one_wave); break;
            default: launch_strip<10>(s, img, k0, k1, mask, mask_pitch, D.max_key, raw, D.blk_count, eig_out_or_null, &g_src, one_wave); break;
        }
    } else if (!use_generic && fused_block_size(block_size)) {
        float k0, k1;
        sobel_scale(block_size, &k0, &k1);
        switch (block_size) {
            case 3: launch_fused<3>(s, img, k0, k1, mask, mask_pitch, D.max_key, raw, D.blk_count, eig_out_or_null, &g_src); break;
            case 5: launch_fused<5>(s, img, k0, k1, mask, mask_pitch, D.max_key, raw, D.blk_count, eig_out_or_null, &g_src); break;
            case 7: launch_fused<7>(s, img, k0, k1, mask, mask_pitch, D.max_key, raw, D.blk_count, eig_out_or_null, &g_src); break;
            default: launch_fused<10>(s, img, k0, k1, mask, mask_pitch, D.max_key, raw, D.blk_count, eig_out_or_null, &g_src); break;
        }
    } else {
        launch_min_eig(s, img, block_size, D.eig, mask, mask_pitch, D.max_key, variant);
        g_src.keys = raw;
        g_src.blk_count = D.blk_count;
        g_src.nblk = 0;
        g_src.region = 256 * NMS_ROWS;
        if (img.w >= 3 && img.h >= 3) {
            dim3 grid((img.w - 2 + 255) / 256, (img.h - 2 + NMS_ROWS - 1) / NMS_ROWS);
            hipLaunchKernelGGL(k_nms_collect, grid, dim3(256), 0, s, D.eig, img.w, img.h, mask, mask_pitch, D.max_key,
                               quality, raw, D.blk_count);
            g_src.nblk = (int)(grid.x * grid.y);
        }
    }
    D.src_nblk = g_src.nblk;
    D.src_region = g_src.region;
}

// minDistance < 1: every candidate above the threshold, flat in D.cand / D.cand_count
void launch_flatten(hipStream_t s, DetectScratch& D, double quality)
{
    hipLaunchKernelGGL(k_flatten, dim3(512), dim3(256), 0, s, src_of(D), D.max_key, quality, D.cand, D.cand_count);
}

// Greedy-equivalent min-distance selection, fully enqueued (no host round trip): candidate regions ->
// D.acc (unsorted accepted keys), D.acc_count; the number of thresholded candidates ends in
// D.cell_start[ncell].  D.undecided[kSuppressLaunches-1] != 0 afterwards means "not converged, call
// continue_min_distance".
// Three launches since round 4 (two before): one detection in a hundred left one or two candidates undecided after the
// second (their blockers were decided in the same launch, after the last poll) and went through the host's tail
// (detect_finish: more launches, a host round trip each, rocPRIM's sort) instead of the device's -- 300 us, and 8 ms the
// first time in a process (profiles/r04_c3_stall.txt).  A third launch picks up what the second left and returns at once
// when that is nothing; the follow-up launches use a small grid (a handful of candidates, every workgroup scans the list).
constexpr int kSuppressLaunches = 3;
static void suppress_launches(hipStream_t s, DetectScratch& D, int w, int h, double min_distance)
{
    const int cell = (int)lrint(min_distance);
    const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
    const double md2 = min_distance * min_distance;
    for (int r = 0; r < kSuppressLaunches; r++)
        hipLaunchKernelGGL(k_suppress, dim3(r < 2 ? 4096 : 512), dim3(CT), 0, s, D.cell_cand, D.cell_start + gw * gh, cell, gw, gh,
                           D.cell_start, D.state, md2, D.undecided, r);
}

static void gather_launch(hipStream_t s, DetectScratch& D, int ncell)
{
    hipLaunchKernelGGL(k_gather_accepted, dim3(1024), dim3(CT), 0, s, D.cell_cand, D.cell_start + ncell, D.state, D.acc,
                       D.acc_count);
}

TailReset tail_reset_of(DetectScratch& D, int ncell)
{
    TailReset r{};
    r.cell_count = D.cell_count;
    r.cell_fill = D.cell_fill;
    r.ncell = ncell;
    r.chunk_tot = D.chunk_tot;
    r.undecided = D.undecided;
    r.acc_count = D.acc_count;
    r.cand_count = D.cand_count;
    r.key_hist = D.key_hist;
    r.prune_key = D.prune_key;
    r.scan_chunk = SCAN_CHUNK;
    r.chunk_stride = CHUNK_TOT_STRIDE;
    r.key_bins = KEY_BINS;
    return r;
}

void launch_min_distance(hipStream_t s, DetectScratch& D, int w, int h, double min_distance, double quality,
                         int prune_want, bool no_gather)
{
    const int cell = (int)lrint(min_distance);
    const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
    const int ncell = gw * gh;
    if (prune_want > 0) {
        hipLaunchKernelGGL(k_key_hist, dim3(256), dim3(CT), 0, s, src_of(D), D.max_key, quality, D.key_hist);
        hipLaunchKernelGGL(k_key_select, dim3(1), dim3(CT), 0, s, D.key_hist, (unsigned)prune_want, D.max_key, D.prune_key);
    }
    hipLaunchKernelGGL(k_cell_count, dim3(4096), dim3(CT), 0, s, src_of(D), D.max_key, quality, D.prune_key, w, cell, gw,
                       D.cell_count, D.chunk_tot);
    hipLaunchKernelGGL(k_scan, dim3((ncell + SCAN_CHUNK - 1) / SCAN_CHUNK), dim3(CT), 0, s, D.cell_count, D.chunk_tot,
                       D.cell_start, ncell);
    hipLaunchKernelGGL(k_cell_fill, dim3(4096), dim3(CT), 0, s, src_of(D), D.max_key, quality, D.prune_key, w, cell, gw,
                       D.cell_start, D.cell_fill, D.cell_cand, D.state);
    suppress_launches(s, D, w, h, min_distance);
    if (!no_gather) gather_launch(s, D, ncell);
}

// more relaxation launches + a fresh gather (only when the first batch left candidates undecided)
void continue_min_distance(hipStream_t s, DetectScratch& D, int w, int h, double min_distance)
{
    const int cell = (int)lrint(min_distance);
    const int ncell = ((w + cell - 1) / cell) * ((h + cell - 1) / cell);
    hipMemsetAsync(D.undecided, 0, sizeof(int) * 8, s);
    hipMemsetAsync(D.acc_count, 0, sizeof(int), s);
    suppress_launches(s, D, w, h, min_distance);
    gather_launch(s, D, ncell);
}

int suppress_launch_count() { return kSuppressLaunches; }

void launch_tail_fused(hipStream_t s, const unsigned long long* keys, int n, float* corners, float* seg_xy,
                       uint8_t* seg_alive, float* seg_tracks, int max_vert, DetectScratch& D, int ncell, int mode)
{
    const int emit_blocks = (n + CT - 1) / CT;
    int blocks = (ncell + CT) / CT;
    if ((mode & 1) && blocks < KEY_BINS / CT) blocks = KEY_BINS / CT;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_tail, dim3(emit_blocks + blocks), dim3(CT), 0, s, keys, n, corners, seg_xy, seg_alive, seg_tracks,
                       max_vert, emit_blocks, D.cell_count, D.cell_fill, ncell, D.chunk_tot, D.undecided, D.acc_count,
                       D.cand_count, D.max_key, D.key_hist, D.prune_key, mode);
}

void launch_emit_corners(hipStream_t s, const unsigned long long* keys, int n, int w, float* xy)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_emit, dim3((n + CT - 1) / CT), dim3(CT), 0, s, keys, n, w, xy);
}

}  // namespace icelk
