// lk_common.h -- device helpers shared by the generic (k_lk.hip) and the specialised (k_lk_fast.hip)
// Lucas-Kanade kernels.  Arithmetic follows SURVEY.md A.6 (OpenCV LKTrackerInvoker, restated).
#pragma once
#include "icelk_internal.h"

namespace icelk {
namespace lk {

constexpr int W_BITS = 14;

__device__ __forceinline__ int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

__device__ __forceinline__ int descale(int v, int n) { return (v + (1 << (n - 1))) >> n; }

struct Weights {
    int w00, w01, w10, w11;
};

// iw = cvRound(frac products * 2^14); cvRound = round-half-even (v_rndne_f32)
__device__ __forceinline__ Weights bilinear_weights(float a, float b)
{
    Weights w;
    w.w00 = (int)rintf((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
    w.w01 = (int)rintf(a * (1.f - b) * (float)(1 << W_BITS));
    w.w10 = (int)rintf((1.f - a) * b * (float)(1 << W_BITS));
    w.w11 = (1 << W_BITS) - w.w00 - w.w01 - w.w10;
    return w;
}

struct TrackResult {
    float x, y;
    float err;
    int status;
    int iters;   // iterations run (measurement)
};

// 64-lane integer sum through the DPP network (no LDS traffic): row prefix sums with row_shr 1/2/4/8,
// then row_bcast15 / row_bcast31 carry the row totals up; lane 63 ends with the wave total, which is
// read back as a scalar.
__device__ __forceinline__ int wave_sum_i32(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);  // row_bcast15 -> rows 1,3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);  // row_bcast31 -> rows 2,3
    return __builtin_amdgcn_readlane(v, 63);
}

// Exact 64-lane sum when G-lane partial sums are known to fit int32 (G = 8 or 16): prefix sums inside
// G-lane groups by DPP, then the 64/G group totals are read back as scalars and added in 64 bits on the
// scalar unit.  Needs |v| * G < 2^31.
template <int G>
__device__ __forceinline__ long long wave_sum_grouped(int v)
{
    static_assert(G == 8 || G == 16, "group size");
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);  // row_shr:4
    if (G == 16) v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);  // row_shr:8
    long long t = 0;
#pragma unroll
    for (int g = 0; g < 64 / G; g++) t += (long long)__builtin_amdgcn_readlane(v, g * G + G - 1);
    return t;
}

// ---- exact 64-bit wave sums through the DPP network alone (round 4) ------------------------------------------------------
// wave_sum_grouped reads the 64/G group totals back with v_readlane and adds them on the scalar unit: 8 readlanes and 24
// scalar instructions (sign extension, add, add-with-carry) per sum at G = 8 -- 28 of the 60 scalar instructions of a
// tracker iteration, on a scalar unit the CU's four SIMDs share.  Here the upper reduction steps stay in the vector unit as
// 64-bit adds: v_add_co_u32 / v_addc_co_u32 take DPP operands like any VOP2, so a step is two instructions, the carry
// travelling through VCC (lanes a step's row mask leaves out execute neither, their VCC bits stay).  Same count of vector
// instructions as before (3 + 1 + 6 + 2 against 3 + 8), one scalar move pair instead of 24 scalar instructions.
// Several sums are reduced side by side: a VGPR written by a VALU instruction may be read through DPP two wait states
// later at the earliest, which the interleaving provides without s_nop (the first step waits explicitly: what the compiler
// scheduled in front of an asm statement is not known to its hazard recogniser).
#define ICELK_DPP64_PAIR(LO, HI, CTRL) \
    "v_add_co_u32_dpp " LO ", vcc, " LO ", " LO " " CTRL "\n\tv_addc_co_u32_dpp " HI ", vcc, " HI ", " HI ", vcc " CTRL "\n\t"
#define ICELK_DPP_SHR8 "row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0"
#define ICELK_DPP_BC15 "row_bcast:15 row_mask:0xa bank_mask:0xf"
#define ICELK_DPP_BC31 "row_bcast:31 row_mask:0xc bank_mask:0xf"

// G = lanes whose int32 sum cannot overflow (16, 8 or 1): those steps run in 32 bits
template <int G>
__device__ __forceinline__ int dpp_low_steps(int v)
{
    static_assert(G == 1 || G == 8 || G == 16, "group size");
    if (G >= 8) {
        v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  // row_shr:1
        v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);  // row_shr:2
        v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);  // row_shr:4
    }
    if (G == 16) v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);  // row_shr:8
    return v;
}

__device__ __forceinline__ long long lane63_i64(int lo, int hi)
{
    const unsigned l = (unsigned)__builtin_amdgcn_readlane(lo, 63);
    const int h = __builtin_amdgcn_readlane(hi, 63);
    return (long long)(((unsigned long long)(unsigned)h << 32) | l);
}

template <int G>
__device__ __forceinline__ void wave_sum2_i64(int v0, int v1, long long& s0, long long& s1)
{
    int l0 = dpp_low_steps<G>(v0), l1 = dpp_low_steps<G>(v1);
    int h0 = l0 >> 31, h1 = l1 >> 31;
    if (G == 1)
        asm("s_nop 1\n\t"
            ICELK_DPP64_PAIR("%0", "%1", "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0") ICELK_DPP64_PAIR("%2", "%3", "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0")
            ICELK_DPP64_PAIR("%0", "%1", "row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0") ICELK_DPP64_PAIR("%2", "%3", "row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0")
            ICELK_DPP64_PAIR("%0", "%1", "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0") ICELK_DPP64_PAIR("%2", "%3", "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0")
            : "+v"(l0), "+v"(h0), "+v"(l1), "+v"(h1) : : "vcc");
    if (G <= 8)
        asm("s_nop 1\n\t" ICELK_DPP64_PAIR("%0", "%1", ICELK_DPP_SHR8) ICELK_DPP64_PAIR("%2", "%3", ICELK_DPP_SHR8)
            : "+v"(l0), "+v"(h0), "+v"(l1), "+v"(h1) : : "vcc");
    asm("s_nop 1\n\t" ICELK_DPP64_PAIR("%0", "%1", ICELK_DPP_BC15) ICELK_DPP64_PAIR("%2", "%3", ICELK_DPP_BC15)
        ICELK_DPP64_PAIR("%0", "%1", ICELK_DPP_BC31) ICELK_DPP64_PAIR("%2", "%3", ICELK_DPP_BC31)
        : "+v"(l0), "+v"(h0), "+v"(l1), "+v"(h1) : : "vcc");
    s0 = lane63_i64(l0, h0);
    s1 = lane63_i64(l1, h1);
}

template <int G>
__device__ __forceinline__ void wave_sum3_i64(int v0, int v1, int v2, long long& s0, long long& s1, long long& s2)
{
    int l0 = dpp_low_steps<G>(v0), l1 = dpp_low_steps<G>(v1), l2 = dpp_low_steps<G>(v2);
    int h0 = l0 >> 31, h1 = l1 >> 31, h2 = l2 >> 31;
    if (G == 1)
        asm("s_nop 1\n\t"
            ICELK_DPP64_PAIR("%0", "%1", "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0") ICELK_DPP64_PAIR("%2", "%3", "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0") ICELK_DPP64_PAIR("%4", "%5", "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0")
            ICELK_DPP64_PAIR("%0", "%1", "row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0") ICELK_DPP64_PAIR("%2", "%3", "row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0") ICELK_DPP64_PAIR("%4", "%5", "row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0")
            ICELK_DPP64_PAIR("%0", "%1", "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0") ICELK_DPP64_PAIR("%2", "%3", "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0") ICELK_DPP64_PAIR("%4", "%5", "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0")
            : "+v"(l0), "+v"(h0), "+v"(l1), "+v"(h1), "+v"(l2), "+v"(h2) : : "vcc");
    if (G <= 8)
        asm("s_nop 1\n\t" ICELK_DPP64_PAIR("%0", "%1", ICELK_DPP_SHR8) ICELK_DPP64_PAIR("%2", "%3", ICELK_DPP_SHR8) ICELK_DPP64_PAIR("%4", "%5", ICELK_DPP_SHR8)
            : "+v"(l0), "+v"(h0), "+v"(l1), "+v"(h1), "+v"(l2), "+v"(h2) : : "vcc");
    asm("s_nop 1\n\t" ICELK_DPP64_PAIR("%0", "%1", ICELK_DPP_BC15) ICELK_DPP64_PAIR("%2", "%3", ICELK_DPP_BC15) ICELK_DPP64_PAIR("%4", "%5", ICELK_DPP_BC15)
        ICELK_DPP64_PAIR("%0", "%1", ICELK_DPP_BC31) ICELK_DPP64_PAIR("%2", "%3", ICELK_DPP_BC31) ICELK_DPP64_PAIR("%4", "%5", ICELK_DPP_BC31)
        : "+v"(l0), "+v"(h0), "+v"(l1), "+v"(h1), "+v"(l2), "+v"(h2) : : "vcc");
    s0 = lane63_i64(l0, h0);
    s1 = lane63_i64(l1, h1);
    s2 = lane63_i64(l2, h2);
}

// correctly rounded int64 -> float for |t| < 2^52, through one exact double
__device__ __forceinline__ float i64_to_float(long long t)
{
    const double d = (double)(int)(t >> 32) * 4294967296.0 + (double)(unsigned)t;
    return (float)d;
}

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uni_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// Exact sum of per-lane int32 partials as int64: the partial is split into a 16-bit low part and a
// signed high part so that neither 64-lane sum can overflow 32 bits.
__device__ __forceinline__ long long wave_sum_exact(int v)
{
    const int lo = wave_sum_i32(v & 0xffff);
    const int hi = wave_sum_i32(v >> 16);
    return ((long long)hi << 16) + (long long)lo;
}

// Workgroup -> feature.  Without an order table workgroup b tracks feature b.  With one, the table is a
// spatially sorted sequence of features and workgroup b takes entry (b % 8) * ceil(count / 8) + b / 8:
// workgroups are dispatched round-robin over the 8 XCDs, so each XCD (own L2) walks one contiguous eighth of
// the sequence.  With order_plain the table is walked linearly (consecutive workgroups = neighbouring features on
// DIFFERENT XCDs).  -1 = nothing to do; the grid must hold count + 8 workgroups (rounded up to 8).
__device__ __forceinline__ int launch_slot(const LKBuffers& B, int b, int count)
{
    if (!B.order) return b < count ? b : -1;
    if (B.order_plain) return b < count ? B.order[b] : -1;
    // the table starts with the border features (slow: launched first, on whatever XCD), the rest is dealt
    const int nb = B.order_border ? *B.order_border : 0;
    if (b < nb) return B.order[b];
    const int rest = count - nb, bb = b - nb;
    const int chunk = (rest + 7) >> 3, j = bb >> 3;
    const int idx = (bb & 7) * chunk + j;
    return (j < chunk && idx < rest) ? B.order[nb + idx] : -1;
}

// segment-mode epilogue of a fused forward+backward launch (one lane per feature calls it)
__device__ __forceinline__ void seg_append(const LKBuffers& B, int f, float x, float y, float d, bool valid)
{
    B.seg_alive[f] = valid ? 1 : 0;
    if (valid) {
        B.seg_xy[2 * f] = x;
        B.seg_xy[2 * f + 1] = y;
        float* t = B.seg_tracks + ((size_t)f * B.seg_max_vert + B.seg_vert) * 2;
        t[0] = x;
        t[1] = y;
        B.seg_quality[(size_t)f * (B.seg_max_vert - 1) + (B.seg_vert - 1)] = d;
    }
    atomicAdd(&B.seg_tracked[f & 63], 1ull);
}

// Forward-backward distance of s1_lucaskanade_tracking.py:329-330: diff = abs(p0 - p0r) in float32, then
// np.hypot on float32 = the C library's hypotf, which glibc evaluates as (float)sqrt((double)x*x + (double)y*y)
// (both products exact in double, one rounding in the sum, one in the correctly rounded sqrt, one in the narrowing).
// form 1 is the demo script's float32 expression (dx**2 + dy**2)**0.5 (s0_1_test_lucaskanade_tracking.py:99).
__device__ __forceinline__ float fb_distance(float p0x, float p0y, float rx, float ry, int form)
{
    const float ddx = fabsf(__fsub_rn(p0x, rx)), ddy = fabsf(__fsub_rn(p0y, ry));
    if (form == 1) return sqrtf(__fadd_rn(__fmul_rn(ddx, ddx), __fmul_rn(ddy, ddy)));
    const double x = (double)ddx, y = (double)ddy;
    return (float)sqrt(__dadd_rn(__dmul_rn(x, x), __dmul_rn(y, y)));
}

// diagnostics: when and where a workgroup ran (lane 0 calls it; B.stamps is null outside diagnostic runs)
__device__ __forceinline__ void stamp(const LKBuffers& B, int which)
{
    if (!B.stamps) return;
    unsigned long long* s = B.stamps + 3 * (size_t)blockIdx.x;
    s[which] = __builtin_amdgcn_s_memtime();
    if (which == 0)
        s[2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
               ((unsigned long long)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15) << 32);
}

template <bool SMALL, int G>
__device__ __forceinline__ long long sum_pick(int v)
{
    if constexpr (SMALL) return wave_sum_grouped<G>(v);
    else return wave_sum_exact(v);
}

}  // namespace lk
}  // namespace icelk
