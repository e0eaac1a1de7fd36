// k_lk_multi.hip -- pyramidal Lucas-Kanade, SEVERAL features per wavefront (gfx950).
//
// Same arithmetic, bit for bit, as k_lk.hip / k_lk_fast.hip (s1_lucaskanade_tracking.py:323,326 and
// the forward-backward test s1:329-333).  What changes is who does the per-feature scalar work.
//
// In k_lk_fast.hip one wave carries one feature, and of the ~155 vector instructions of an iteration only ~65 touch
// pixels: the rest -- bilinear weights, the two exact sums, the 2x2 solve, the convergence tests -- is computed on
// wave-uniform values, i.e. the same number in all 64 lanes, plus ~70 scalar instructions of 64-bit adds.  The VALU
// issue slot is what bounds the tracker (profiles/r02_valu_rate.txt: 4.3-5 cycles per wave64 instruction of the
// dot2 / perm / mad kind), so that uniform work is the waste.  Here a wave carries F features (4 for 21x21) in
// lockstep:
//   * PIXEL PHASES (once per feature): lane = row segment of the window, as before; the feature's uniform operands
//     (packed weights, tile offset) come from the owning lanes by v_readlane.  Per-lane partial sums go to LDS
//     unreduced.
//   * UNIFORM PHASES (once per wave): the 64 lanes are split into F blocks of 64/F lanes, block f owns feature f.
//     Every lane of the block keeps the feature's state (position, 2x2 matrix, flags) in its own registers; the
//     exact sums are finished here: each lane takes F of the 64 partials of its feature, splits them in 16-bit
//     halves (no overflow possible), and an all-reduce by DPP row rotations leaves both half sums in every lane of
//     the block: float(hi) * 65536 + float(lo) is then the correctly rounded float of the exact integer sum, as
//     the oracle's (float)int64.  No 64-bit arithmetic, no scalar add chains, no double precision.
//   * the template of a feature (I as a dot2 accumulator seed, Ix / Iy as packed 16-bit pairs of neighbouring
//     pixels) stays in registers across the iterations: 15 VGPRs per feature at 21x21; residual * gradient sums are
//     v_dot2_i32_i16 on packed residual pairs.
//   * double precision is gone from the loop: (double)dx*dx + (double)dy*dy <= eps^2 is decided in float whenever
//     the float value is outside a 2^-20 band around eps^2 (otherwise the exact form runs), and
//     fabs((double)t) < 0.01 for a float t is |t| <= 0.01f (0.01f is the largest float below 0.01).
//   * all global loads of a level -- template patch and search tile of all F features -- are in flight together.
// Features of a wave that finish a level early simply sit out the remaining pixel phases (scalar branches).
#include <stdlib.h>

#include "lk_fast_tiles.h"

namespace icelk {

namespace {

using namespace lk;
using namespace lkf;

__device__ __forceinline__ int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ uint32_t rlu(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }

// sum over the 64 / F lanes of a block, result in every lane of the block.  16-lane rows: rotations by 8, 4, 2, 1
// (DPP row_ror); blocks of 32 / 64 lanes add the partner rows through ds_swizzle / ds_bpermute.
template <int F>
__device__ __forceinline__ int block_allreduce(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);  // row_ror:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false);  // row_ror:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x122, 0xf, 0xf, false);  // row_ror:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, false);  // row_ror:1
    if constexpr (F <= 2) v += __shfl_xor(v, 16);
    if constexpr (F == 1) v += __shfl_xor(v, 32);
    return v;
}

// The exact integer sum of the 64 per-lane partials of this lane's feature, as a correctly rounded float.
// sums[] holds F x 64 int32 partials (feature-major); lane l owns dwords [l*F, l*F + F).
template <int F>
__device__ __forceinline__ float block_sum_f32(const uint32_t* sums, int lane)
{
    int lo = 0, hi = 0;
    if constexpr (F == 4) {
        const uint4 v = *reinterpret_cast<const uint4*>(sums + 4 * lane);
        lo = (int)((v.x & 0xffffu) + (v.y & 0xffffu) + (v.z & 0xffffu) + (v.w & 0xffffu));
        hi = ((int)v.x >> 16) + ((int)v.y >> 16) + ((int)v.z >> 16) + ((int)v.w >> 16);
    } else if constexpr (F == 2) {
        const uint2 v = *reinterpret_cast<const uint2*>(sums + 2 * lane);
        lo = (int)((v.x & 0xffffu) + (v.y & 0xffffu));
        hi = ((int)v.x >> 16) + ((int)v.y >> 16);
    } else {
        const uint32_t v = sums[lane];
        lo = (int)(v & 0xffffu);
        hi = (int)v >> 16;
    }
    lo = block_allreduce<F>(lo);   // <= 64 * 65535 < 2^24
    hi = block_allreduce<F>(hi);   // |.| <= 64 * 32768 = 2^21
    // both converts are exact, the product is exact, so the one rounding of the add is the rounding of the exact sum
    return __fadd_rn(__fmul_rn((float)hi, 65536.f), (float)lo);
}

template <int WW, int WH, int F>
struct MCfg {
    using C = Cfg<WW, WH>;
    static constexpr int LPF = 64 / F;                 // lanes per feature in the uniform phases
    static constexpr int NP2 = (C::S + 1) / 2;         // packed gradient pairs per row segment
    static constexpr int TILE_DW = C::I_DW + C::J_DW;  // LDS dwords of one feature's two tiles
    static constexpr int SUMS_DW = 3 * F * 64;         // three sums in flight at most (template phase)
    static constexpr int LDS_DW = F * TILE_DW + SUMS_DW + 8;
};

// per-feature state kept by every lane of the feature's block
struct FeatState {
    float p0x, p0y;     // the point being tracked (level 0 coordinates)
    float sx, sy;       // the stored nextPts value
    float err;
    int status;
    int iters;          // iterations run so far (measurement)
    bool present;       // the block has a feature at all
};

// stage the search tile of feature FI at (x0, y0) (scalars)
template <int WW, int WH>
__device__ __forceinline__ void stage_j(uint32_t* ldsJ, const Level& LJ, int x0, int y0, int lane)
{
    using C = Cfg<WW, WH>;
    if (tile_inside(LJ, x0, y0, C::JTW, C::JTH)) {
        TileRegs<C::JPD, C::JTH> tj;
        tile_issue(tj, LJ, x0, y0, lane);
        tile_commit(tj, ldsJ, lane);
    } else {
        tile_border<C::JPD, C::JTW, C::JTH>(ldsJ, LJ, x0, y0, lane);
    }
}


// One direction of the pyramidal tracker for the F features of this wave.  S.sx / S.sy / S.status / S.err are the results.
template <int WW, int WH, int F>
__device__ __forceinline__ void track_multi(const Pyramid& PI, const Pyramid& PJ, FeatState& St, const LKParams& P,
                                            uint32_t* lds, int lane)
{
    using C = Cfg<WW, WH>;
    using M = MCfg<WW, WH, F>;
    constexpr int S = C::S;
    constexpr int R = kMargin;
    constexpr int LPF = M::LPF;
    const float half_x = (WW - 1) * 0.5f, half_y = (WH - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    uint32_t* const sums = lds + F * M::TILE_DW;
    constexpr unsigned long long kOwner = F == 4 ? 0x0001000100010001ull : (F == 2 ? 0x0000000100000001ull : 1ull);

    // this lane's row segments (pixel phases)
    int trow[C::TPL], tcol[C::TPL], tlen[C::TPL], joff[C::TPL];
#pragma unroll
    for (int k = 0; k < C::TPL; k++) {
        const int t = lane + 64 * k;
        const bool used = t < C::NTASK;   // surplus lanes work on segment 0 with every pixel masked off
        const int row = used ? t / C::NSEG : 0;
        trow[k] = row;
        tcol[k] = used ? (t - row * C::NSEG) * S : 0;
        tlen[k] = used ? (WW - tcol[k] < S ? WW - tcol[k] : S) : 0;
        joff[k] = row * C::JPD * 4 + tcol[k];
    }
    // which of a segment's pixels belong to the window, as masks over the packed gradient pairs
    uint32_t pmask[C::TPL][(S + 1) / 2];
#pragma unroll
    for (int k = 0; k < C::TPL; k++)
#pragma unroll
        for (int q = 0; q < (S + 1) / 2; q++)
            pmask[k][q] = (2 * q < tlen[k] ? 0xffffu : 0u) | (2 * q + 1 < tlen[k] ? 0xffff0000u : 0u);

    St.status = 1;
    St.err = 0.f;
    Template<WW, WH, F> T;

    for (int level = P.top_level; level >= 0; level--) {
        const Level LI = PI.lv[level];
        const Level LJ = PJ.lv[level];
        // the template of a level is used inside that level only; telling the compiler that the registers hold nothing
        // here (an empty asm that "writes" them) keeps them out of the live set of the staging code below
        static_for<F>([&](auto ff) {
            constexpr int f = ff;
#pragma unroll
            for (int k = 0; k < C::TPL; k++) {
#pragma unroll
                for (int q = 0; q < S; q++) asm volatile("" : "=v"(T.Ineg[f][k][q]));
#pragma unroll
                for (int q = 0; q < (S + 1) / 2; q++) {
                    asm volatile("" : "=v"(T.Ixp[f][k][q]));
                    asm volatile("" : "=v"(T.Iyp[f][k][q]));
                }
            }
        });
        // ---- uniform phase: where is the template, where does the search start ---------------------------------------
        const float scale = 1.f / (float)(1 << level);
        float px = St.p0x * scale, py = St.p0y * scale;
        if (level == P.top_level) { St.sx = px; St.sy = py; }
        else { St.sx = St.sx * 2.f; St.sy = St.sy * 2.f; }
        px -= half_x; py -= half_y;
        const int ipx = (int)floorf(px), ipy = (int)floorf(py);
        bool lvl = St.present && origin_ok<WW, WH>(LI, ipx, ipy);
        if (St.present && !lvl && level == 0) { St.status = 0; St.err = 0.f; }
        const Weights wi = bilinear_weights(px - (float)ipx, py - (float)ipy);
        const uint32_t W0u = pack_weights_lo(wi), W1u = pack_weights_hi(wi);
        float nx = St.sx - half_x, ny = St.sy - half_y;
        const int ix0 = ipx - 1, iy0 = ipy - 1;
        const bool i_in = tile_inside(LI, ix0, iy0, C::ITW, C::ITH);
        const int inx0 = (int)floorf(nx), iny0 = (int)floorf(ny);
        const bool j_ok = origin_ok<WW, WH>(LJ, inx0, iny0);
        int jx0 = inx0 - R, jy0 = iny0 - R;
        const bool j_in = j_ok && tile_inside(LJ, jx0, jy0, C::JTW, C::JTH);
        bool staged = j_ok;
        const unsigned long long m_lvl = __builtin_amdgcn_ballot_w64(lvl) & kOwner;
        if (m_lvl == 0) continue;
        const unsigned long long m_iin = __builtin_amdgcn_ballot_w64(i_in), m_jin = __builtin_amdgcn_ballot_w64(j_in),
                                 m_jok = __builtin_amdgcn_ballot_w64(j_ok);

        // ---- stage the template source patch and (speculatively) the first search tile of every feature --------------
        __syncthreads();
        {
            TileRegs<C::IPD, C::ITH> ti[F];
            TileRegs<C::JPD, C::JTH> tj[F];
            // tiles over the frame border take the reflecting loader; either way every load of the level is in flight
            // before the first LDS write waits for one
            static_for<F>([&](auto ff) {
                constexpr int f = ff, o = f * LPF;
                if ((m_lvl >> o) & 1) {
                    if ((m_iin >> o) & 1) tile_issue(ti[f], LI, rl(ix0, o), rl(iy0, o), lane);
                    else tile_issue_reflect(ti[f], LI, rl(ix0, o), rl(iy0, o), lane);
                    if ((m_jin >> o) & 1) tile_issue(tj[f], LJ, rl(jx0, o), rl(jy0, o), lane);
                    else if ((m_jok >> o) & 1) tile_issue_reflect(tj[f], LJ, rl(jx0, o), rl(jy0, o), lane);
                }
            });
            static_for<F>([&](auto ff) {
                constexpr int f = ff, o = f * LPF;
                uint32_t* ldsI = lds + f * M::TILE_DW;
                uint32_t* ldsJ = ldsI + C::I_DW;
                if ((m_lvl >> o) & 1) {
                    tile_commit(ti[f], ldsI, lane);
                    if ((m_jok >> o) & 1) tile_commit(tj[f], ldsJ, lane);
                }
            });
        }
        __syncthreads();

        // ---- pixel phases: template patches ---------------------------------------------------------------------------
        static_for<F>([&](auto ff) {
            constexpr int f = ff, o = f * LPF;
            if ((m_lvl >> o) & 1) {
                int a11, a12, a22;
                template_pixels<WW, WH, F, f>(T, lds + f * M::TILE_DW, rlu(W0u, o), rlu(W1u, o), rl(ix0, o) & 3,
                                              (m_iin >> o) & 1, rl(ipx, o), rl(ipy, o), LI.w, LI.h, trow, tcol, pmask, a11,
                                              a12, a22);
                sums[0 * F * 64 + f * 64 + lane] = (uint32_t)a11;
                sums[1 * F * 64 + f * 64 + lane] = (uint32_t)a12;
                sums[2 * F * 64 + f * 64 + lane] = (uint32_t)a22;
            }
        });
        __syncthreads();

        // ---- uniform phase: 2x2 matrix, its smaller eigenvalue, its inverse determinant ---------------------------------
        const float A11 = block_sum_f32<F>(sums + 0 * F * 64, lane) * FLT_SCALE;
        const float A12 = block_sum_f32<F>(sums + 1 * F * 64, lane) * FLT_SCALE;
        const float A22 = block_sum_f32<F>(sums + 2 * F * 64, lane) * FLT_SCALE;
        float D = __fsub_rn(__fmul_rn(A11, A22), __fmul_rn(A12, A12));
        {
            const float dif = __fsub_rn(A11, A22);
            const float rad = __fadd_rn(__fmul_rn(dif, dif), __fmul_rn(__fmul_rn(4.f, A12), A12));
            const float minEig = __fdiv_rn(__fsub_rn(__fadd_rn(A22, A11), sqrtf(rad)), (float)(2 * WW * WH));
            if (lvl) {
                if (P.flags & ICELK_FLAG_MIN_EIGENVALS) St.err = minEig;
                if (minEig < P.min_eig_thr || D < 1.1920928955078125e-07f) {
                    if (level == 0) St.status = 0;
                    lvl = false;
                }
            }
        }
        D = __fdiv_rn(1.f, D);

        // ---- iterations, all features of the wave in lockstep -------------------------------------------------------------
        bool act = lvl;
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < P.max_count; j++) {
            // uniform phase: window origin of this iteration
            int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (act && !origin_ok<WW, WH>(LJ, inx, iny)) {
                if (level == 0) St.status = 0;
                act = false;
            }
            unsigned long long m_act = __builtin_amdgcn_ballot_w64(act) & kOwner;
            if (m_act == 0) break;
            St.iters += act ? 1 : 0;
            const bool need = act && !(staged && tile_covers(jx0, jy0, inx, iny));
            const unsigned long long m_need = __builtin_amdgcn_ballot_w64(need) & kOwner;
            if (need) { jx0 = inx - R; jy0 = iny - R; staged = true; }
            if (m_need) {
                __syncthreads();
                for (unsigned long long mr = m_need; mr; mr &= mr - 1) {
                    const int o = __builtin_ctzll(mr);
                    stage_j<WW, WH>(lds + (o / LPF) * M::TILE_DW + C::I_DW, LJ, rl(jx0, o), rl(jy0, o), lane);
                }
                __syncthreads();
            }
            const Weights wj = bilinear_weights(nx - (float)inx, ny - (float)iny);
            const uint32_t V0u = pack_weights_lo(wj), V1u = pack_weights_hi(wj);
            const int jb = (iny - jy0) * (C::JPD * 4) + (jx0 & 3) + (inx - jx0);

            // pixel phases
            static_for<F>([&](auto ff) {
                constexpr int f = ff, o = f * LPF;
                if ((m_act >> o) & 1) {
                    int b1, b2;
                    residual_pixels<WW, WH, F, f, false>(T, lds + f * M::TILE_DW + C::I_DW, rl(jb, o), rlu(V0u, o),
                                                         rlu(V1u, o), joff, tlen, b1, b2);
                    sums[0 * F * 64 + f * 64 + lane] = (uint32_t)b1;
                    sums[1 * F * 64 + f * 64 + lane] = (uint32_t)b2;
                }
            });
            __syncthreads();

            // uniform phase: solve, move, test
            const float fb1 = block_sum_f32<F>(sums + 0 * F * 64, lane) * FLT_SCALE;
            const float fb2 = block_sum_f32<F>(sums + 1 * F * 64, lane) * FLT_SCALE;
            __syncthreads();   // the sums are consumed: the next pixel phase may overwrite them
            const float dx = __fmul_rn(__fsub_rn(__fmul_rn(A12, fb2), __fmul_rn(A22, fb1)), D);
            const float dy = __fmul_rn(__fsub_rn(__fmul_rn(A12, fb1), __fmul_rn(A11, fb2)), D);
            if (act) {
                nx = __fadd_rn(nx, dx); ny = __fadd_rn(ny, dy);
                St.sx = __fadd_rn(nx, half_x); St.sy = __fadd_rn(ny, half_y);
            }
            // (double)dx*dx + (double)dy*dy <= eps^2: the float value decides unless it lies in the 2^-20 band around eps^2
            const float q = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
            bool conv = q < P.eps2_lo;
            const bool unsure = act && !conv && !(q > P.eps2_hi);
            if (__builtin_amdgcn_ballot_w64(unsure) != 0)
                conv = __dadd_rn(__dmul_rn((double)dx, (double)dx), __dmul_rn((double)dy, (double)dy)) <= P.eps2;
            if (act) {
                if (conv) {
                    act = false;
                } else if (j > 0 && fabsf(__fadd_rn(dx, pdx)) <= 0.01f && fabsf(__fadd_rn(dy, pdy)) <= 0.01f) {
                    // fabs((double)t) < 0.01 for a float t  <=>  |t| <= 0.01f, the largest float below 0.01
                    St.sx = __fsub_rn(St.sx, __fmul_rn(dx, 0.5f));
                    St.sy = __fsub_rn(St.sy, __fmul_rn(dy, 0.5f));
                    act = false;
                } else {
                    pdx = dx; pdy = dy;
                }
            }
        }

        // ---- residual error at level 0 ----------------------------------------------------------------------------------------
        if (level == 0 && !(P.flags & ICELK_FLAG_MIN_EIGENVALS)) {
            bool want = lvl && St.status != 0;
            const float qx = St.sx - half_x, qy = St.sy - half_y;
            const int iqx = (int)floorf(qx), iqy = (int)floorf(qy);
            if (want && !origin_ok<WW, WH>(LJ, iqx, iqy)) {
                St.status = 0;
                want = false;
            }
            const unsigned long long m_want = __builtin_amdgcn_ballot_w64(want) & kOwner;
            if (m_want != 0) {
                const bool need = want && !(staged && tile_covers(jx0, jy0, iqx, iqy));
                const unsigned long long m_need = __builtin_amdgcn_ballot_w64(need) & kOwner;
                if (need) { jx0 = iqx - R; jy0 = iqy - R; staged = true; }
                if (m_need) {
                    __syncthreads();
                    for (unsigned long long mr = m_need; mr; mr &= mr - 1) {
                        const int o = __builtin_ctzll(mr);
                        stage_j<WW, WH>(lds + (o / LPF) * M::TILE_DW + C::I_DW, LJ, rl(jx0, o), rl(jy0, o), lane);
                    }
                    __syncthreads();
                }
                const Weights we = bilinear_weights(qx - (float)iqx, qy - (float)iqy);
                const uint32_t V0u = pack_weights_lo(we), V1u = pack_weights_hi(we);
                const int jb = (iqy - jy0) * (C::JPD * 4) + (jx0 & 3) + (iqx - jx0);
                static_for<F>([&](auto ff) {
                    constexpr int f = ff, o = f * LPF;
                    if ((m_want >> o) & 1) {
                        int es, unused;
                        residual_pixels<WW, WH, F, f, true>(T, lds + f * M::TILE_DW + C::I_DW, rl(jb, o), rlu(V0u, o),
                                                            rlu(V1u, o), joff, tlen, es, unused);
                        sums[f * 64 + lane] = (uint32_t)es;
                    }
                });
                __syncthreads();
                const float errval = block_sum_f32<F>(sums, lane);
                __syncthreads();
                if (want) St.err = __fdiv_rn(__fmul_rn(errval, 1.f), (float)(32 * WW * WH));
            }
        }
    }
}

// Workgroup b, feature block f -> feature.  Without an order table workgroup b tracks features b*F .. b*F+F-1.  With
// one, the table is a spatially sorted sequence; dealt to the XCDs in contiguous eighths (workgroups are dispatched
// round-robin over the 8 XCDs), F consecutive entries per workgroup -- neighbours in the frame share L2 lines.
template <int F>
__device__ __forceinline__ int feature_slot(const LKBuffers& B, int b, int f, int count)
{
    if (!B.order || B.order_plain) {
        const int i = b * F + f;
        if (i >= count) return -1;
        return B.order ? B.order[i] : i;
    }
    // border features first (F per workgroup), the rest dealt to the XCDs in contiguous eighths
    const int nb = B.order_border ? *B.order_border : 0;
    const int nbw = (nb + F - 1) / F;   // workgroups that carry border features
    if (b < nbw) {
        const int i = b * F + f;
        return i < nb ? B.order[i] : -1;
    }
    const int rest = count - nb, bb = b - nbw;
    const int chunk = (rest + 7) >> 3, j = (bb >> 3) * F + f;
    const int idx = (bb & 7) * chunk + j;
    return (j < chunk && idx < rest) ? B.order[nb + idx] : -1;
}

template <int WW, int WH, int F>
constexpr int waves_per_simd()
{
    // LDS: 160 KB per CU, 4 SIMDs
    constexpr int by_lds = (160 * 1024) / (MCfg<WW, WH, F>::LDS_DW * 4) / 4;
    return by_lds >= 4 ? 4 : (by_lds >= 3 ? 3 : (by_lds >= 2 ? 2 : 1));
}

template <int WW, int WH, int F, bool FB>
__global__ __launch_bounds__(64, (F == 1 && Cfg<WW, WH>::TPL == 1) ? 4 : 1) void k_lk_multi(Pyramid PI, Pyramid PJ, LKBuffers B, int n, LKParams P)
{
    using M = MCfg<WW, WH, F>;
    constexpr int LPF = M::LPF;
    __shared__ __attribute__((aligned(16))) uint32_t lds[M::LDS_DW];
    const int lane = threadIdx.x;
    const int fi = lane / LPF;
    const int count = B.n_dev ? *B.n_dev : n;
    const int slot = feature_slot<F>(B, blockIdx.x, fi, count);
    FeatState St;
    St.present = slot >= 0;
    if (St.present && B.seg_alive) St.present = B.seg_alive[slot] != 0;
    if (__builtin_amdgcn_ballot_w64(St.present) == 0) return;
    if (lane == 0) stamp(B, 0);
    St.p0x = St.present ? B.p_in[2 * slot] : 0.f;
    St.p0y = St.present ? B.p_in[2 * slot + 1] : 0.f;
    const float p0x = St.p0x, p0y = St.p0y;
    const bool writer = St.present && (lane % LPF) == 0;
    St.iters = 0;
    track_multi<WW, WH, F>(PI, PJ, St, P, lds, lane);
    const float fx = St.sx, fy = St.sy;
    const int it_fwd = St.iters;
    if (writer && B.iters && !FB) B.iters[slot] = (uint32_t)it_fwd;
    if (writer) {
        if (B.p_fwd) { B.p_fwd[2 * slot] = fx; B.p_fwd[2 * slot + 1] = fy; }
        if (B.st_fwd) B.st_fwd[slot] = (uint8_t)St.status;
        if (B.err_fwd) B.err_fwd[slot] = St.err;
    }
    if (FB) {
        St.p0x = fx;
        St.p0y = fy;
        St.iters = 0;
        track_multi<WW, WH, F>(PJ, PI, St, P, lds, lane);
        if (writer) {
            if (B.iters) B.iters[slot] = (uint32_t)it_fwd | ((uint32_t)St.iters << 16);
            if (B.p_bwd) { B.p_bwd[2 * slot] = St.sx; B.p_bwd[2 * slot + 1] = St.sy; }
            if (B.st_bwd) B.st_bwd[slot] = (uint8_t)St.status;
            if (B.err_bwd) B.err_bwd[slot] = St.err;
            const float d = fb_distance(p0x, p0y, St.sx, St.sy, P.dist_form);
            if (B.dist) B.dist[slot] = d;
            if (B.valid) B.valid[slot] = d < P.fb_thr ? 1 : 0;
            if (B.seg_alive) seg_append(B, slot, fx, fy, d, d < P.fb_thr);
        }
    }
    if (lane == 0) stamp(B, 1);
}

template <int WW, int WH, int F>
void launch_multi(hipStream_t s, const Pyramid& I, const Pyramid& J, const LKBuffers& B, int n, const LKParams& P, bool fb)
{
    int grid;
    if (B.order && !B.order_plain) {
        // ceil(nb / F) border workgroups + 8 * ceil(ceil((n - nb) / 8) / F) dealt ones <= n / F + 17 for any nb
        grid = n / F + 18;
    } else {
        grid = (n + F - 1) / F;
    }
    if (fb) hipLaunchKernelGGL((k_lk_multi<WW, WH, F, true>), dim3(grid), dim3(64), 0, s, I, J, B, n, P);
    else hipLaunchKernelGGL((k_lk_multi<WW, WH, F, false>), dim3(grid), dim3(64), 0, s, I, J, B, n, P);
}

}  // namespace

// Returns true when a multi-feature kernel exists for this window (and INITIAL_FLOW is not requested).
bool launch_lk_multi(hipStream_t s, const Pyramid& I, const Pyramid& J, const LKBuffers& B, int n, const LKParams& P,
                     bool fb)
{
    if (P.flags & ICELK_FLAG_INITIAL_FLOW) return false;
    static const int exp_f = getenv("ICELK_LK_F") ? atoi(getenv("ICELK_LK_F")) : 0;   // experiments: features per wave
    if (P.win_w == 21 && P.win_h == 21 && exp_f == 2) launch_multi<21, 21, 2>(s, I, J, B, n, P, fb);
    else if (P.win_w == 21 && P.win_h == 21 && exp_f == 1) launch_multi<21, 21, 1>(s, I, J, B, n, P, fb);
    else if (P.win_w == 21 && P.win_h == 21) launch_multi<21, 21, 4>(s, I, J, B, n, P, fb);
    else if (P.win_w == 15 && P.win_h == 15) launch_multi<15, 15, 4>(s, I, J, B, n, P, fb);
    else if (P.win_w == 31 && P.win_h == 31) launch_multi<31, 31, 2>(s, I, J, B, n, P, fb);
    else if (P.win_w == 35 && P.win_h == 35) launch_multi<35, 35, 2>(s, I, J, B, n, P, fb);
    else return false;
    return true;
}

}  // namespace icelk
