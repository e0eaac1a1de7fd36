// k_corners_fast.hip -- Shi-Tomasi corner candidates in two passes: an INTEGER map that brackets OpenCV's float map
// from both sides, then OpenCV's exact float arithmetic only where a corner can be.
//
// A third statement of the candidate stage (K6 + K7) of cv2.goodFeaturesToTrack(frame_gray, mask=mask, **feature_params)
// at s1_lucaskanade_tracking.py:437 for blockSize 3 / 5 / 7 / 10, selected with ICELK_TWO_PASS_CORNERS=1.  Its result --
// per-region lists of 64-bit keys (response key << 32 | y << 16 | x) of every local maximum, and the masked maximum of the
// map -- is bit for bit what the one-pass kernel k_eig_nms (k_corners.hip) produces (tests/test_gpu_parity.py::
// test_two_pass_corner_detector, and the oracle).  NOT the default: measured at C2 it is slower inside the pipeline
// (DESIGN.md 4.2 has the numbers and the reason), because a textured frame has 2.6 * 10^5 local maxima above
// max * qualityLevel that all need their exact value before the top-K pruning of the min-distance stage can drop 2/3 of
// them.  It is kept because the bracket is the tool for that next step (pruning on the bounds, before the exact pass).
//
// Idea.  k_eig_nms evaluates OpenCV's float pipeline at every pixel: 243 lane-operations per pixel at blockSize 10 (Sobel
// in float, three planes of double-precision box sums with a 9-px halo, a correctly rounded sqrtf) -- for a map of which
// only the local maxima are ever looked at.  Here:
//
//   pass A  k_eig_approx   for every pixel the structure-tensor sums of the UNSCALED integer Sobel derivatives,
//                          Sxx = sum dxi^2, Sxy = sum dxi dyi, Syy = sum dyi^2 over the blockSize window: exact in int32
//                          (|dxi| <= 1020, 100 terms: 27 bits), on packed 16-bit math and v_dot2_i32_i16.  From them
//                          L = (Sxx+Syy)/2 - sqrt(((Sxx-Syy)/2)^2 + Sxy^2) in f32 and a bound eps(p) such that OpenCV's
//                          float value v(p) satisfies  k1^2 (L - eps) <= v(p) <= k1^2 (L + eps)   [bound derived below].
//                          Listed per tile: every pixel that CAN be a local maximum (L + eps >= the largest L - eps of
//                          its 3x3 neighbourhood, L + eps > 0), flagged CERTAIN when L - eps > 0 and > L + eps of each
//                          of its eight neighbours; and every pixel that can carry the masked maximum.
//                          141 lane-operations per pixel (26.9 M wave instructions at 12 MP), 65 us alone.
//   pass B0 k_exact_max    OpenCV's float arithmetic at the handful of pixels that can carry the maximum -> max_key.
//   pass B1 k_exact_cands  listed pixels whose upper bound stays below max * qualityLevel are dropped; the certain ones
//                          get OpenCV's float value at the pixel itself (a quad of lanes each) and, if > 0, their key;
//                          the rest -- possible ties, 1 % -- go to pass B2.
//   pass B2 k_exact_ties   OpenCV's float values on the 3x3 neighbourhood (nine quads side by side), v > 0 and v >= its
//                          eight neighbours decided on the exact floats.
//
// Error bound (all in "integer units": v / k1^2 with k1 = (float)(1 / (4 blockSize 255)), k0 = 2 k1 exactly; u = 2^-24).
// OpenCV (and oracle/icelk_oracle.c: orc_min_eig_map) forms
//     dx = fl(fl((r0 + r2) k1) + fl(r1 k0)),  r = right - left pixel (exact);   dy = fl(t2 - t0),
//     t = fl(fl(fl(k1 a) + fl(k0 b)) + fl(k1 c))  on pixels a, b, c in [0, 255].
//   |dx - k1 dxi| <= 2.0001 u k1 (|r0 + r2| + 2 |r1|) <= k1 ex,  ex = 2040.2 u = 1.216e-4
//   |t - k1 (a + 2b + c)| <= 3.0001 u k1 1020;  |dy - k1 dyi| <= k1 ey,  ey = (2 * 3060.1 + 1020) u = 4.256e-4
//   products: |fl(dx dx) - k1^2 dxi^2| <= k1^2 (2 ex |dxi| + ex^2 + u (|dxi| + ex)^2), likewise the other two;
//   box sums are exact in double (k_corners.hip), so with N = blockSize^2, T = Sxx + Syy and Cauchy-Schwarz
//   sum |dxi| <= sqrt(N Sxx) <= sqrt(N T):
//     |Sxx_f - k1^2 Sxx| <= k1^2 (2 ex sqrt(N T) + N ex^2 + u T)   (yy with ey;  xy: (ex + ey) sqrt(N T) + N ex ey + u T / 2)
//   the smaller eigenvalue of a symmetric 2x2 matrix moves by at most the spectral norm of a perturbation,
//   <= max(|da|, |dc|) + |db| with a = Sxx/2, c = Syy/2, b = Sxy; the float evaluation (float)s0 * 0.5f ... (a + c) - sqrtf(..)
//   adds <= 3.4 u T on top (every rounding is relative to a quantity <= (a + c), the sqrt halves its argument's).  Together
//     |v / k1^2 - L| <= (ex + 2 ey) sqrt(N T) + N (ey^2 / 2 + ex ey) + 4.4 u T = 9.73e-4 sqrt(N T) + 1.43e-7 N + 2.62e-7 T.
//   Pass A's own f32 evaluation of L from the exact integers (two int -> float conversions, an approximate sqrt) stays below
//   3.6 u T = 2.1e-7 T.  The kernels use  eps = 1.25e-3 sqrt(N T) + 1e-6 T + 1e-6 N  (>= 1.28 / 2.1 / 7 times the terms), and
//   T = 0 (all derivatives of the window zero) means v = 0 exactly.  The tests compare the result with the one-pass kernel
//   and with the oracle on every image they use, incl. the full BASELINE frames (tests/test_gpu_fullsize_oracle.py).
#include "lk_fast_tiles.h"

namespace icelk {

namespace {

using namespace lkf;

constexpr int FT_TW = 62, FT_TH = 32;          // output pixels per tile
constexpr int FT_EW = 64, FT_EH = FT_TH + 2;   // eigenvalue region: outputs + a 1-px ring (one wave lane per column)
constexpr int FT_RY = 8;                       // output rows per wave of the column pass (4 waves)
constexpr int FT_SEG = 16;                     // row-sum columns per row task
constexpr int FT_CAP = 256;                    // possible local maxima listed per tile (more: the tile is walked whole)
constexpr int FT_KMAX = 16;                    // possible carriers of the maximum listed per tile
constexpr int FT_G = 4;                        // tiles per wave of the exact pass
static_assert(FT_TH == 4 * FT_RY, "four waves cover the tile");

constexpr float EPS_SQRT = 1.25e-3f, EPS_LIN = 1.0e-6f, EPS_N = 1.0e-6f;

__device__ __forceinline__ int reflect101c(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// BORDER_REFLECT_101 runs through an image of n samples with period 2n - 2; on the odd half-periods it runs backwards.
// A derivative formed on the reflected SAMPLES at such a position is the negative of the derivative AT the reflected
// position (the smoothing taps are symmetric, so the other derivative is unchanged) -- and OpenCV's box filter reflects
// the covariance image, i.e. takes the derivatives at the reflected position.
__device__ __forceinline__ bool reflect_backwards(int p, int n)
{
    if (n <= 1) return false;
    const int period = 2 * n - 2;
    int q = p % period;
    if (q < 0) q += period;
    return q > n - 1;
}

__device__ __forceinline__ unsigned ordered_key_f(float v)
{
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_to_float_f(unsigned k)
{
    const unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(b);
}

__device__ __forceinline__ void atomic_max_guarded(unsigned* p, unsigned v)
{
    if (v > __atomic_load_n(p, __ATOMIC_RELAXED)) atomicMax(p, v);
}

// 64-lane maximum through the DPP network (row_shr 1/2/4/8, row_bcast15/31); lane 63 ends with it, read back as a scalar.
// The operands are ordered INTEGER keys (no float canonicalisation on the way).
__device__ __forceinline__ int wave_max_i32(int v)
{
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, 63);
}
// float -> int whose signed order is the float order
__device__ __forceinline__ int float_order(float f)
{
    const int b = __float_as_int(f);
    return b ^ ((b >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float order_float(int k) { return __int_as_float(k ^ ((k >> 31) & 0x7fffffff)); }

template <int BS>
struct ACfg {
    static constexpr int AN = BS / 2;
    static constexpr int HR = FT_EH + BS - 1;           // rows of row sums
    static constexpr int UW = FT_EW + BS + 1, UH = HR + 2;   // u8 region
    static constexpr int UPD = (UW + 3) / 4 + 1;        // its LDS row pitch in dwords (any 4-byte phase)
    static constexpr int NPC = FT_SEG + BS - 1;         // product columns a row task forms
    static constexpr int NPAIR = (NPC + 1) / 2;
    static constexpr int NXD = (2 * (NPAIR + 1) + 3) / 4;   // u8 dwords of one source row of a task
    static constexpr int U_BYTES = (UPD * UH * 4 + 4 * NXD + 15) & ~15;   // + slack: realignment reads run past a row
    static constexpr int HP = FT_EW + 4;                // row pitch of the row sums: a wave's 16-B writes spread over all banks
    static constexpr int HS_INTS = 3 * HR * HP;
    static constexpr int LDS_BYTES = U_BYTES + HS_INTS * 4;
    static constexpr int NROWTASK = HR * (FT_EW / FT_SEG);
    static_assert(NROWTASK <= 256, "one row task per thread");
};

// One row task of pass A: the integer Sobel pair of NPC consecutive positions of one row from three rows of the u8 region,
// prefix sums of the three products along the row, the FT_SEG window sums -> hs.  BORDER: the tile reaches over the frame
// border and holds reflected samples; where the reflection runs backwards in exactly one of x, y the product dxi dyi changes
// sign (see reflect_backwards) -- applied to dxi, whose square does not care.
template <int BS, bool BORDER>
__device__ __forceinline__ void row_task(const uint32_t* __restrict__ U, int* __restrict__ hs, int cs, int hrow, int g, int x0,
                                         int y0, int w, int h)
{
    using C = ACfg<BS>;
    constexpr int NP1 = C::NPAIR + 1;
    v2s P[3][NP1];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        uint32_t X[C::NXD];
        row_dwords<C::NXD>(U + (hrow + r) * C::UPD, cs + FT_SEG * g, X);
        static_for<NP1>([&](auto jj) { P[r][jj] = byte_pair<2 * jj, C::NXD>(X); });
    }
    v2s S[NP1], Dv[NP1];
#pragma unroll
    for (int j = 0; j < NP1; j++) {
        S[j] = (P[0][j] + P[2][j]) + (P[1][j] + P[1][j]);   // [1 2 1] down the rows: <= 1020
        Dv[j] = P[2][j] - P[0][j];                           // bottom - top
    }
    unsigned flips = 0;   // bit k: product column k of this task changes the sign of dxi dyi
    if (BORDER) {
        const bool fy = reflect_backwards(y0 - 1 - C::AN + hrow, h);
        const int xp0 = x0 - 1 - C::AN + FT_SEG * g;
#pragma unroll 1
        for (int k = 0; k < 2 * C::NPAIR; k++) flips |= (reflect_backwards(xp0 + k, w) != fy ? 1u : 0u) << k;
    }
    int Pxx[C::NPC], Pxy[C::NPC], Pyy[C::NPC];
    int pxx = 0, pxy = 0, pyy = 0;
    static_for<C::NPAIR>([&](auto jj) {
        constexpr int j = jj;
        v2s DX = S[j + 1] - S[j];                                                  // dxi at product columns 2j, 2j+1
        const v2s DY = (Dv[j] + Dv[j + 1]) + (pair_shift(Dv[j], Dv[j + 1]) + pair_shift(Dv[j], Dv[j + 1]));
        if (BORDER) {
            const uint32_t f = ((flips >> (2 * j)) & 1u ? 0xffffu : 0u) | ((flips >> (2 * j + 1)) & 1u ? 0xffff0000u : 0u);
            DX = as_v2s(as_u32(DX) ^ f) - as_v2s(f);     // -v = (v ^ 0xffff) - 0xffff per 16-bit half
        }
        // prefix sums of the products: column 2j from the low halves alone, column 2j+1 from the whole pair
        const v2s dxl = as_v2s(as_u32(DX) & 0xffffu), dyl = as_v2s(as_u32(DY) & 0xffffu);
        Pxx[2 * j] = dot2(DX, dxl, pxx); Pxy[2 * j] = dot2(DY, dxl, pxy); Pyy[2 * j] = dot2(DY, dyl, pyy);
        if constexpr (2 * j + 1 < C::NPC) {
            pxx = dot2(DX, DX, pxx); pxy = dot2(DY, DX, pxy); pyy = dot2(DY, DY, pyy);
            Pxx[2 * j + 1] = pxx; Pxy[2 * j + 1] = pxy; Pyy[2 * j + 1] = pyy;
        }
    });
    int* o = hs + hrow * C::HP + FT_SEG * g;
#pragma unroll
    for (int q = 0; q < FT_SEG / 4; q++) {
        int4 a, b, c;
        int* ap = reinterpret_cast<int*>(&a);
        int* bp = reinterpret_cast<int*>(&b);
        int* cp = reinterpret_cast<int*>(&c);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int col = 4 * q + k;
            ap[k] = Pxx[col + BS - 1] - (col ? Pxx[col ? col - 1 : 0] : 0);
            bp[k] = Pxy[col + BS - 1] - (col ? Pxy[col ? col - 1 : 0] : 0);
            cp[k] = Pyy[col + BS - 1] - (col ? Pyy[col ? col - 1 : 0] : 0);
        }
        *reinterpret_cast<int4*>(o + 4 * q) = a;
        *reinterpret_cast<int4*>(o + C::HR * C::HP + 4 * q) = b;
        *reinterpret_cast<int4*>(o + 2 * C::HR * C::HP + 4 * q) = c;
    }
}

// ---- pass A ------------------------------------------------------------------------------------------------------------
template <int BS>
__global__ __launch_bounds__(256) void k_eig_approx(const uint8_t* __restrict__ img, int w, int h, int pitch,
                                                    const uint8_t* __restrict__ mask, int mask_pitch,
                                                    unsigned* __restrict__ max_key, unsigned* __restrict__ fmax_key,
                                                    uint2* __restrict__ acand, int* __restrict__ acount,
                                                    uint2* __restrict__ amaxc, int* __restrict__ amaxn,
                                                    float* __restrict__ aemax)
{
    using C = ACfg<BS>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t* U = reinterpret_cast<uint32_t*>(smem);
    int* hs = reinterpret_cast<int*>(smem + C::U_BYTES);
    __shared__ int s_ncand, s_nmax;
    __shared__ float s_wF[4], s_wE[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int x0 = blockIdx.x * FT_TW, y0 = blockIdx.y * FT_TH;
    const int ux0 = x0 - 2 - C::AN, uy0 = y0 - 2 - C::AN;
    const int cs = ux0 & 3;
    const bool interior = ux0 >= 0 && uy0 >= 0 && ux0 + C::UW <= w && uy0 + C::UH <= h;
    if (tid == 0) { s_ncand = 0; s_nmax = 0; }

    if (interior) {
        const uint8_t* base = img + (size_t)uy0 * pitch + (ux0 & ~3);
        for (int i = tid; i < C::UPD * C::UH; i += 256) {
            const int r = i / C::UPD, c = i - r * C::UPD;
            U[i] = *reinterpret_cast<const uint32_t*>(base + (size_t)r * pitch + 4 * c);
        }
    } else {
        uint8_t* Ub = reinterpret_cast<uint8_t*>(U);
        for (int i = tid; i < C::UW * C::UH; i += 256) {
            const int r = i / C::UW, c = i - r * C::UW;
            Ub[r * C::UPD * 4 + cs + c] = img[(size_t)reflect101c(uy0 + r, h) * pitch + reflect101c(ux0 + c, w)];
        }
    }
    __syncthreads();

    // ---- row pass: integer Sobel of one row segment, products, window sums along the row --------------------------------
    if (tid < C::NROWTASK) {
        const int hrow = tid / (FT_EW / FT_SEG), g = tid - hrow * (FT_EW / FT_SEG);
        if (interior) row_task<BS, false>(U, hs, cs, hrow, g, x0, y0, w, h);
        else row_task<BS, true>(U, hs, cs, hrow, g, x0, y0, w, h);
    }
    __syncthreads();

    // ---- column pass: lane = eigenvalue column, wave = a band of FT_RY output rows + its ring ----------------------------
    constexpr int NE = FT_RY + 2;            // eigenvalue rows of a band
    constexpr int NR = NE + BS - 1;          // row-sum rows it reads
    float E[NE], F[NE];
    unsigned zero_bits = 0;                  // bit i: T == 0 at eigenvalue row i (v is exactly 0 there)
    {
        const int* hp = hs + (FT_RY * wv) * C::HP + lane;
        int sum[3][NE];
#pragma unroll
        for (int p = 0; p < 3; p++) {
            int R[NR];
#pragma unroll
            for (int k = 0; k < NR; k++) R[k] = hp[p * C::HR * C::HP + k * C::HP];
            int s = R[0];
#pragma unroll
            for (int k = 1; k < BS; k++) s += R[k];
            sum[p][0] = s;
#pragma unroll
            for (int i = 1; i < NE; i++) {
                s += R[i + BS - 1] - R[i - 1];
                sum[p][i] = s;
            }
        }
        constexpr float fN = (float)(BS * BS);
#pragma unroll
        for (int i = 0; i < NE; i++) {
            const int T = sum[0][i] + sum[2][i], Dd = sum[0][i] - sum[2][i];
            const float fT = (float)T, fD = (float)Dd, fXY = (float)sum[1][i];
            const float arg = fmaf(4.f * fXY, fXY, fD * fD);
            const float L2 = fT - __builtin_amdgcn_sqrtf(arg);                 // 2 L
            float eps = fmaf(EPS_SQRT, __builtin_amdgcn_sqrtf(fN * fT), fmaf(EPS_LIN, fT, EPS_N * fN));
            if (T == 0) { eps = 0.f; zero_bits |= 1u << i; }
            E[i] = fmaf(0.5f, L2, eps);
            F[i] = fmaf(0.5f, L2, -eps);
        }
    }
    const int x = x0 - 1 + lane;
    const bool col_out = lane >= 1 && lane <= FT_TW && x < w;
    const int tile = blockIdx.y * gridDim.x + blockIdx.x;
    const int yb = y0 - 1 + FT_RY * wv;     // image row of eigenvalue row 0 of this band
    // which of the band's output rows are pixels of the image inside the mask (rows below the frame: none)
    unsigned live = 0;
    if (col_out) {
        const int nrow = h - yb - 1 < FT_RY ? h - yb - 1 : FT_RY;      // output rows i = 1 .. nrow are inside the frame
        live = nrow > 0 ? ((2u << nrow) - 2u) : 0u;
        if (mask) {
#pragma unroll
            for (int i = 1; i <= FT_RY; i++)
                if ((live & (1u << i)) && !mask[(size_t)(yb + i) * mask_pitch + x]) live &= ~(1u << i);
        }
    }
    // possible local maxima: upper bound >= the largest lower bound of the 3x3 neighbourhood (its own included), > 0
    // ... and of those the CERTAIN ones: lower bound > 0 and > the upper bound of each of the eight neighbours -- the exact
    // value at the pixel itself is then all the exact pass has to form
    unsigned cand = 0, sure = 0;
    float fbest = -INFINITY, ebest = -INFINITY;
#pragma unroll
    for (int i = 1; i <= FT_RY; i++) {
        const float cm3 = fmaxf(fmaxf(F[i - 1], F[i]), F[i + 1]);
        float m9 = fmaxf(cm3, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(cm3), 0x138, 0xf, 0xf, false)));   // wave_shr:1
        m9 = fmaxf(m9, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(cm3), 0x130, 0xf, 0xf, false)));         // wave_shl:1
        cand |= (E[i] > 0.f && E[i] >= m9) ? 1u << i : 0u;
        const float ce3 = fmaxf(fmaxf(E[i - 1], E[i]), E[i + 1]);
        float e8 = fmaxf(E[i - 1], E[i + 1]);
        e8 = fmaxf(e8, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ce3), 0x138, 0xf, 0xf, false)));
        e8 = fmaxf(e8, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ce3), 0x130, 0xf, 0xf, false)));
        sure |= (F[i] > 0.f && F[i] > e8) ? 1u << i : 0u;
        const bool inexact = ((live & ~zero_bits) >> i) & 1u;
        fbest = inexact ? fmaxf(fbest, F[i]) : fbest;
        ebest = inexact ? fmaxf(ebest, E[i]) : ebest;
    }
    {
        // the 1-px frame border holds no corner (x and the rows as one mask)
        const int lo = 1 - yb > 1 ? 1 - yb : 1, hi = h - 2 - yb < FT_RY ? h - 2 - yb : FT_RY;   // rows i with 1 <= y <= h-2
        const unsigned rows_ok = hi >= lo ? ((2u << hi) - (1u << lo)) : 0u;
        cand &= live & rows_ok & ((x >= 1 && x <= w - 2) ? ~0u : 0u);
    }
    while (cand) {
        const int i = __ffs((int)cand) - 1;
        cand &= cand - 1;
        float e = E[1];
#pragma unroll
        for (int k = 2; k <= FT_RY; k++) e = i == k ? E[k] : e;
        if ((sure >> i) & 1u) e = -e;          // the sign carries the flag (the bound itself is > 0)
        const int pos = atomicAdd(&s_ncand, 1);
        if (pos < FT_CAP) acand[(size_t)tile * FT_CAP + pos] = make_uint2(((unsigned)(yb + i) << 16) | (unsigned)x, __float_as_uint(e));
    }
    // exactly-zero pixels publish their exact value themselves; the inexact ones a lower bound of the maximum
    if (__builtin_amdgcn_ballot_w64((live & zero_bits) != 0u) != 0 && lane == 0) atomic_max_guarded(max_key, ordered_key_f(0.f));
    const float wf = order_float(wave_max_i32(float_order(fbest))), we = order_float(wave_max_i32(float_order(ebest)));
    if (lane == 0) { s_wF[wv] = wf; s_wE[wv] = we; }
    __syncthreads();
    const float tileF = fmaxf(fmaxf(s_wF[0], s_wF[1]), fmaxf(s_wF[2], s_wF[3]));
    const float tileE = fmaxf(fmaxf(s_wE[0], s_wE[1]), fmaxf(s_wE[2], s_wE[3]));
    if (ebest >= tileF) {      // the few lanes whose best pixel reaches the tile's largest lower bound
        unsigned mc = live & ~zero_bits;
        while (mc) {
            const int i = __ffs((int)mc) - 1;
            mc &= mc - 1;
            float e = E[1];
#pragma unroll
            for (int k = 2; k <= FT_RY; k++) e = i == k ? E[k] : e;
            if (e >= tileF) {
                const int pos = atomicAdd(&s_nmax, 1);
                if (pos < FT_KMAX) amaxc[(size_t)tile * FT_KMAX + pos] = make_uint2(((unsigned)(yb + i) << 16) | (unsigned)x, __float_as_uint(e));
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        acount[tile] = s_ncand;
        amaxn[tile] = s_nmax;
        aemax[tile] = tileE;
        if (tileF > -INFINITY) atomic_max_guarded(fmax_key, ordered_key_f(tileF));
    }
}

// ---- OpenCV's float arithmetic at single pixels ------------------------------------------------------------------------
__device__ __forceinline__ float min_eig_exact(double s0, double s1, double s2)
{
    const float a = __fmul_rn((float)s0, 0.5f), b = (float)s1, c = __fmul_rn((float)s2, 0.5f);
    const float d = __fsub_rn(a, c);
    return __fsub_rn(__fadd_rn(a, c), sqrtf(__fadd_rn(__fmul_rn(d, d), __fmul_rn(b, b))));
}

// The same value at ONE pixel by FOUR neighbouring lanes (a quad): each forms a quarter of the window's derivative rows and
// their box-sum share, the three double sums are added across the quad (exact in double: any order) -- a quarter of the
// dependent instruction chain per lane, ~100 registers, and four times as many waves to hide the load latency with.
// Every lane of the quad returns the value.
__device__ __forceinline__ double quad_sum(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    v += __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xf, 0xf, false),
                          __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    lo = __double2loint(v); hi = __double2hiint(v);
    v += __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xf, 0xf, false),
                          __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    return v;
}

template <int BS>
__device__ __forceinline__ float exact_eig_quad(const uint8_t* __restrict__ img, int w, int h, int pitch, int x, int y, float k0,
                                                float k1, int q)
{
    constexpr int AN = BS / 2;
    constexpr int PW = BS + 2;
    constexpr int ND = (PW + 3) / 4 + 1;
    constexpr int NQ = (BS + 3) / 4;                 // derivative rows per lane, at most
    constexpr int base = BS / 4, extra = BS % 4;
    const int n = base + (q < extra ? 1 : 0);        // this lane's rows: m0 .. m0 + n - 1
    const int m0 = q * base + (q < extra ? q : extra);
    const int px0 = x - AN - 1, py0 = y - AN - 1;
    double acc[3] = {0.0, 0.0, 0.0};
    auto add_row = [&](bool live, const float (&dx)[BS], const float (&dy)[BS]) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int i = 0; i < BS; i++) {
            s0 += (double)__fmul_rn(dx[i], dx[i]);
            s1 += (double)__fmul_rn(dx[i], dy[i]);
            s2 += (double)__fmul_rn(dy[i], dy[i]);
        }
        if (live) { acc[0] += s0; acc[1] += s1; acc[2] += s2; }
    };
    if (px0 >= 0 && py0 >= 0 && px0 + PW <= w && py0 + PW <= h) {
        const uint8_t* base_p = img + (size_t)py0 * pitch + (px0 & ~3);
        const int sh = px0 & 3;
        uint32_t raw[NQ + 2][ND];
#pragma unroll
        for (int j = 0; j < NQ + 2; j++) {
            const int row = m0 + j < PW ? m0 + j : PW - 1;       // a lane with fewer rows re-reads the last one
#pragma unroll
            for (int d = 0; d < ND; d++) raw[j][d] = *reinterpret_cast<const uint32_t*>(base_p + (size_t)row * pitch + 4 * d);
        }
        float rdx[3][BS], rdy[3][BS];
#pragma unroll
        for (int j = 0; j < NQ + 2; j++) {
            uint32_t al[ND - 1];
#pragma unroll
            for (int d = 0; d < ND - 1; d++) al[d] = __builtin_amdgcn_alignbyte(raw[j][d + 1], raw[j][d], sh);
            float pix[PW];
#pragma unroll
            for (int i = 0; i < PW; i++) pix[i] = (float)((al[i / 4] >> (8 * (i % 4))) & 255u);
#pragma unroll
            for (int i = 0; i < BS; i++) {
                rdx[j % 3][i] = __fsub_rn(pix[i + 2], pix[i]);
                rdy[j % 3][i] = __fadd_rn(__fadd_rn(__fmul_rn(k1, pix[i]), __fmul_rn(k0, pix[i + 1])), __fmul_rn(k1, pix[i + 2]));
            }
            if (j < 2) continue;
            const int mm = j - 2, r0 = mm % 3, r1 = (mm + 1) % 3, r2 = (mm + 2) % 3;
            float dx[BS], dy[BS];
#pragma unroll
            for (int i = 0; i < BS; i++) {
                dx[i] = __fadd_rn(__fmul_rn(__fadd_rn(rdx[r0][i], rdx[r2][i]), k1), __fmul_rn(rdx[r1][i], k0));
                dy[i] = __fsub_rn(rdy[r2][i], rdy[r0][i]);
            }
            add_row(mm < n, dx, dy);
        }
    } else {
#pragma unroll 1
        for (int mm = 0; mm < n; mm++) {
            const int ry = reflect101c(py0 + 1 + m0 + mm, h);
            const uint8_t* r0 = img + (size_t)reflect101c(ry - 1, h) * pitch;
            const uint8_t* r1 = img + (size_t)ry * pitch;
            const uint8_t* r2 = img + (size_t)reflect101c(ry + 1, h) * pitch;
            float dx[BS], dy[BS];
#pragma unroll
            for (int i = 0; i < BS; i++) {
                const int rx = reflect101c(px0 + 1 + i, w);
                const int xm = reflect101c(rx - 1, w), xp = reflect101c(rx + 1, w);
                const float a0 = (float)r0[xm], b0 = (float)r0[rx], c0 = (float)r0[xp];
                const float a1 = (float)r1[xm], c1 = (float)r1[xp];
                const float a2 = (float)r2[xm], b2 = (float)r2[rx], c2 = (float)r2[xp];
                dx[i] = __fadd_rn(__fmul_rn(__fadd_rn(__fsub_rn(c0, a0), __fsub_rn(c2, a2)), k1), __fmul_rn(__fsub_rn(c1, a1), k0));
                const float t0 = __fadd_rn(__fadd_rn(__fmul_rn(k1, a0), __fmul_rn(k0, b0)), __fmul_rn(k1, c0));
                const float t2 = __fadd_rn(__fadd_rn(__fmul_rn(k1, a2), __fmul_rn(k0, b2)), __fmul_rn(k1, c2));
                dy[i] = __fsub_rn(t2, t0);
            }
            add_row(true, dx, dy);
        }
    }
    return min_eig_exact(quad_sum(acc[0]), quad_sum(acc[1]), quad_sum(acc[2]));
}

// ---- pass B0: the masked maximum of the map ----------------------------------------------------------------------------
template <int BS>
__global__ __launch_bounds__(64) void k_exact_max(const uint8_t* __restrict__ img, int w, int h, int pitch, float k0, float k1,
                                                  const uint8_t* __restrict__ mask, int mask_pitch, int ntiles, int tiles_x,
                                                  const unsigned* __restrict__ fmax_key, const uint2* __restrict__ amaxc,
                                                  const int* __restrict__ amaxn, const float* __restrict__ aemax,
                                                  unsigned* __restrict__ max_key)
{
    const unsigned fk = *fmax_key;
    if (fk == 0u) return;                        // no inexact pixel anywhere: the maximum is already exact (or nothing is masked in)
    const float Fg = key_to_float_f(fk);
    // a quad of lanes per list entry (exact_eig_quad); every condition below is the same for the four lanes of a quad
    const int gid = blockIdx.x * 64 + threadIdx.x;
    const int ent = gid >> 2, q = gid & 3;
    const int tile = ent / FT_KMAX, slot = ent - tile * FT_KMAX;
    if (tile >= ntiles) return;
    const int n = amaxn[tile];
    unsigned best = 0;
    if (n <= FT_KMAX) {
        if (slot < n) {
            const uint2 e = amaxc[(size_t)tile * FT_KMAX + slot];
            if (__uint_as_float(e.y) >= Fg)
                best = ordered_key_f(exact_eig_quad<BS>(img, w, h, pitch, (int)(e.x & 0xffffu), (int)(e.x >> 16), k0, k1, q));
        }
    } else if (aemax[tile] >= Fg) {
        // more pixels of this tile tie for its maximum than the list holds (a plateau): walk the tile
        const int tx = tile % tiles_x, ty = tile / tiles_x;
        for (int p = slot; p < FT_TW * FT_TH; p += FT_KMAX) {
            const int x = tx * FT_TW + p % FT_TW, y = ty * FT_TH + p / FT_TW;
            if (x >= w || y >= h || (mask && !mask[(size_t)y * mask_pitch + x])) continue;
            const unsigned k = ordered_key_f(exact_eig_quad<BS>(img, w, h, pitch, x, y, k0, k1, q));
            best = k > best ? k : best;
        }
    }
    if (best && q == 0) atomic_max_guarded(max_key, best);
}

// ---- pass B1: exact keys of the certain local maxima; the others are handed on ------------------------------------------
// The lists of FT_G tiles per wave.  First every listed pixel is looked at by one lane: upper bound below max * qualityLevel
// -> dropped (two thirds of a textured frame's local maxima); CERTAIN (pass A: strictly above all eight neighbours whatever
// the rounding) -> kept in LDS; the rest -- possible ties -- and tiles whose list overflowed go to pass B2's list.  Then the
// kept pixels get OpenCV's float value at the pixel itself, a quad of lanes each (exact_eig_quad), and -- if it is > 0 --
// their key.
template <int BS>
__global__ __launch_bounds__(64) void k_exact_cands(const uint8_t* __restrict__ img, int w, int h, int pitch, float k0, float k1,
                                                    int ntiles, const unsigned* __restrict__ max_key, double quality,
                                                    const uint2* __restrict__ acand, const int* __restrict__ acount,
                                                    unsigned long long* __restrict__ raw, int* __restrict__ blk_count,
                                                    unsigned* __restrict__ ties, int* __restrict__ tie_count)
{
    __shared__ unsigned kept[FT_G * FT_CAP];
    const int g = blockIdx.x, lane = threadIdx.x;
    // v <= k1^2 (L + eps) (1 + 2^-22): nothing below the quality threshold needs the exact arithmetic (quality <= 0: no cut)
    const unsigned mk = *max_key;
    const double max_val = mk ? (double)key_to_float_f(mk) : 0.0;
    const float thr_up = quality > 0 ? (float)((double)(float)(max_val * quality) / ((double)k1 * (double)k1 * 1.000001)) * 0.999999f
                                     : -1.f;     // in pass A's units, rounded down
    int start[FT_G + 1];
    start[0] = 0;
#pragma unroll
    for (int t = 0; t < FT_G; t++) {
        const int tile = g * FT_G + t;
        int n = tile < ntiles ? acount[tile] : 0;
        if (n > FT_CAP) {
            // the list overflowed: pass B2 walks the whole tile (entry = tile | 0x80000000)
            if (lane == 0) ties[atomicAdd(tie_count, 1)] = 0x80000000u | (unsigned)tile;
            n = 0;
        }
        start[t + 1] = start[t] + n;
    }
    int nkept = 0;
    for (int base = 0; base < start[FT_G]; base += 64) {
        const int idx = base + lane;
        bool sure = false, tie = false;
        unsigned xy = 0;
        if (idx < start[FT_G]) {
            int t = 0;
#pragma unroll
            for (int k = 1; k < FT_G; k++) t += idx >= start[k] ? 1 : 0;
            const uint2 e = acand[(size_t)(g * FT_G + t) * FT_CAP + (idx - start[t])];
            xy = e.x;
            const float eb = __uint_as_float(e.y);
            const bool above = fabsf(eb) > thr_up;
            sure = above && eb < 0.f;
            tie = above && !(eb < 0.f);
        }
        const unsigned long long tb = __builtin_amdgcn_ballot_w64(tie);
        if (tb) {   // possible ties: one aggregated append per wave
            int tbase = 0;
            if (lane == 0) tbase = atomicAdd(tie_count, __popcll(tb));
            tbase = __shfl(tbase, 0);
            if (tie) ties[tbase + __popcll(tb & ((1ull << lane) - 1ull))] = xy;
        }
        const unsigned long long sb = __builtin_amdgcn_ballot_w64(sure);
        if (sure) kept[nkept + __popcll(sb & ((1ull << lane) - 1ull))] = xy;
        nkept += __popcll(sb);
    }
    __syncthreads();
    if (lane == 0 && nkept) atomicAdd(tie_count + 1, nkept);     // work counter (icelk_detect_fast_stats)
    unsigned long long* region = raw + (size_t)g * (FT_G * FT_TW * FT_TH);
    const int q = lane & 3;
    int written = 0;
    for (int base = 0; base < nkept; base += 16) {
        const int idx = base + (lane >> 2);
        unsigned long long key = 0;
        bool keep = false;
        if (idx < nkept) {
            const unsigned xy = kept[idx];
            const float v = exact_eig_quad<BS>(img, w, h, pitch, (int)(xy & 0xffffu), (int)(xy >> 16), k0, k1, q);
            keep = v > 0.f && q == 0;
            key = ((unsigned long long)ordered_key_f(v) << 32) | xy;
        }
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(keep);
        if (keep) region[written + __popcll(bal & ((1ull << lane) - 1ull))] = key;
        written += __popcll(bal);
    }
    if (lane == 0) blk_count[g] = written;
}

// ---- pass B2: possible ties -- OpenCV's float values on the 3x3 neighbourhood, v > 0 and v >= the eight others ------------
// A wave per entry at a time: nine quads of lanes form the nine values side by side (one quad evaluation deep instead of nine).
template <int BS>
__global__ __launch_bounds__(64) void k_exact_ties(const uint8_t* __restrict__ img, int w, int h, int pitch, float k0, float k1,
                                                   const uint8_t* __restrict__ mask, int mask_pitch, int tiles_x,
                                                   const unsigned* __restrict__ ties, const int* __restrict__ tie_count,
                                                   unsigned long long* __restrict__ raw, int* __restrict__ blk_count)
{
    const int n = *tie_count, lane = threadIdx.x;
    const int k9 = lane >> 2, q = lane & 3;                 // quad k9 < 9 forms the value at (x + k9 % 3 - 1, y + k9 / 3 - 1)
    auto settle = [&](int x, int y) {
        float v = -INFINITY;
        if (k9 < 9) v = exact_eig_quad<BS>(img, w, h, pitch, x + k9 % 3 - 1, y + k9 / 3 - 1, k0, k1, q);
        const float centre = __shfl(v, 16);
        const float m = order_float(wave_max_i32(float_order(k9 == 4 ? -INFINITY : v)));
        if (lane == 0 && centre > 0.f && !(centre < m)) {
            const int g = ((y / FT_TH) * tiles_x + x / FT_TW) / FT_G;
            raw[(size_t)g * (FT_G * FT_TW * FT_TH) + atomicAdd(&blk_count[g], 1)] =
                ((unsigned long long)ordered_key_f(centre) << 32) | ((unsigned)y << 16) | (unsigned)x;
        }
    };
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const unsigned e = ties[i];
        if (!(e & 0x80000000u)) {
            settle((int)(e & 0xffffu), (int)(e >> 16));
            continue;
        }
        // a whole tile (its list overflowed in pass A): every pixel that may hold a corner
        const int tile = (int)(e & 0x7fffffffu), tx = tile % tiles_x, ty = tile / tiles_x;
        for (int p = 0; p < FT_TW * FT_TH; p++) {
            const int x = tx * FT_TW + p % FT_TW, y = ty * FT_TH + p / FT_TW;
            if (x >= 1 && x <= w - 2 && y >= 1 && y <= h - 2 && (!mask || mask[(size_t)y * mask_pitch + x])) settle(x, y);
        }
    }
}

void sobel_scale_f(int block_size, float* k0, float* k1)
{
    double scale = (double)(1 << 2) * block_size;
    scale *= 255.0;
    scale = 1.0 / scale;
    *k1 = (float)(1.0 * scale);
    *k0 = (float)(2.0 * scale);
}

template <int BS>
void launch_fast(hipStream_t s, const DetectScratch& D, const Level& img, const uint8_t* mask, int mask_pitch, double quality,
                 int* nblk, int* region)
{
    using C = ACfg<BS>;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_eig_approx<BS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            C::LDS_BYTES);
        attr_set = true;
    }
    float k0, k1;
    sobel_scale_f(BS, &k0, &k1);
    const int tx = (img.w + FT_TW - 1) / FT_TW, ty = (img.h + FT_TH - 1) / FT_TH;
    const int ntiles = tx * ty, ngroups = (ntiles + FT_G - 1) / FT_G;
    hipMemsetAsync(D.fmax_key, 0, 3 * sizeof(unsigned), s);   // and the tie / kept counters behind it
    hipLaunchKernelGGL((k_eig_approx<BS>), dim3(tx, ty), dim3(256), C::LDS_BYTES, s, img.ptr, img.w, img.h, img.pitch, mask,
                       mask_pitch, D.max_key, D.fmax_key, D.acand, D.acount, D.amaxc, D.amaxn, D.aemax);
    hipLaunchKernelGGL((k_exact_max<BS>), dim3((ntiles * FT_KMAX * 4 + 63) / 64), dim3(64), 0, s, img.ptr, img.w, img.h, img.pitch, k0,
                       k1, mask, mask_pitch, ntiles, tx, D.fmax_key, D.amaxc, D.amaxn, D.aemax, D.max_key);
    int* tie_count = reinterpret_cast<int*>(D.fmax_key + 1);
    hipLaunchKernelGGL((k_exact_cands<BS>), dim3(ngroups), dim3(64), 0, s, img.ptr, img.w, img.h, img.pitch, k0, k1, ntiles,
                       D.max_key, quality, D.acand, D.acount, D.raw, D.blk_count, D.aties, tie_count);
    hipLaunchKernelGGL((k_exact_ties<BS>), dim3(2048), dim3(64), 0, s, img.ptr, img.w, img.h, img.pitch, k0, k1, mask, mask_pitch,
                       tx, D.aties, tie_count, D.raw, D.blk_count);
    *nblk = ngroups;
    *region = FT_G * FT_TW * FT_TH;
}

}  // namespace

// scratch the two-pass detector needs per candidate buffer, for a w x h frame
size_t fast_tiles(int w, int h) { return (size_t)((w + FT_TW - 1) / FT_TW) * ((h + FT_TH - 1) / FT_TH); }
size_t fast_cand_entries(int w, int h) { return fast_tiles(w, h) * FT_CAP; }
size_t fast_max_entries(int w, int h) { return fast_tiles(w, h) * FT_KMAX; }
// keys / regions the exact pass writes (same raw buffer and count array the one-pass kernel uses)
size_t fast_key_capacity(int w, int h) { return ((fast_tiles(w, h) + FT_G - 1) / FT_G) * (size_t)(FT_G * FT_TW * FT_TH); }
size_t fast_regions(int w, int h) { return (fast_tiles(w, h) + FT_G - 1) / FT_G; }

// K6 + K7 in two passes (see the head of this file); D.max_key must have been zeroed by the caller.  quality <= 0: every
// local maximum gets its exact key (a later detection with any qualityLevel can adopt the result).
bool launch_candidates_fast(hipStream_t s, DetectScratch& D, const Level& img, int block_size, const uint8_t* mask,
                            int mask_pitch, double quality)
{
    int nblk = 0, region = 0;
    switch (block_size) {
        case 3: launch_fast<3>(s, D, img, mask, mask_pitch, quality, &nblk, &region); break;
        case 5: launch_fast<5>(s, D, img, mask, mask_pitch, quality, &nblk, &region); break;
        case 7: launch_fast<7>(s, D, img, mask, mask_pitch, quality, &nblk, &region); break;
        case 10: launch_fast<10>(s, D, img, mask, mask_pitch, quality, &nblk, &region); break;
        default: return false;
    }
    D.src_nblk = nblk;
    D.src_region = region;
    return true;
}

}  // namespace icelk
