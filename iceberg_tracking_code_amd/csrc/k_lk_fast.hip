// k_lk_fast.hip -- window-size-specialised pyramidal Lucas-Kanade for gfx950 (21x21, 31x31, 35x35, ...).
//
// Same arithmetic, bit for bit, as the generic kernel in k_lk.hip (which stays the path for every other
// winSize); this file is the MI355X-tuned form for the window sizes that BASELINE.json and the reference
// (s1_lucaskanade_tracking.py:246, winSize=(35,35)) use.  One wavefront per feature, all levels and both
// directions of the forward-backward check in one launch.
//
// What the compile-time window buys:
//   * lane = one ROW SEGMENT of the window (S <= 8 contiguous pixels of one window row): 21x21 -> 63
//     segments of 7 px, one per lane.  A lane reads its pixels from LDS as aligned dwords and realigns
//     them with v_alignbyte, 6 ds_read_b32 per iteration instead of 28 ds_read_u8.
//   * pixels are widened to 16-bit PAIRS (v_perm_b32) and every bilinear tap pair is one
//     v_dot2_i32_i16 against a packed weight pair (w00|w01, w10|w11): a sample costs 2 dot2 + 1 shift;
//     the rounding constant and the template value ride in the dot2 accumulator.
//   * the Scharr derivative of the template is formed in registers with packed 16-bit math
//     (v_pk_add/mad/sub_u16, two columns per instruction) from the 4 source rows a segment touches;
//     nothing but the u8 source patch and the u8 search tile ever sits in LDS.
//   * the VALU issue slot (one wave64 instruction per 4 cycles per SIMD) is what bounds this kernel,
//     so the design goal is instructions per sample, not bytes.
//   * tiles are staged as aligned dwords, all global loads of a level (template patch AND search tile)
//     issued before the first one is waited for; divisions by the tile pitch are by constants.
//   * the five sums are exact integers reduced through the DPP network (lk_common.h), no LDS round trips.
#include "lk_fast_tiles.h"

namespace icelk {

namespace {

using namespace lk;
using namespace lkf;

// exact int64 -> float for |t| < 2^47: float(t >> 16) * 65536 and float(t & 0xffff) are exact, so the one rounding of
// their sum is the rounding of t itself (what the oracle's (float)int64 does) -- no double precision involved
__device__ __forceinline__ float sum_to_float(long long t)
{
    const int hi = (int)(t >> 16), lo = (int)(t & 0xffff);
    return __fadd_rn(__fmul_rn((float)hi, 65536.f), (float)lo);
}

// A template as it travels from the backward pass of one pair to the forward pass of the next (LKBuffers::tmpl_out):
// the lane's registers, flattened to dwords -- Ineg, then the packed gradient pairs -- in 16-byte pieces.
template <int WW, int WH>
struct TmplIO {
    using C = Cfg<WW, WH>;
    static constexpr int HP = (C::S + 1) / 2;
    // The window values travel as the 13-bit samples they were made from (iv, 1/32 grey levels; Ineg = 256 - (iv << 9) is
    // formed again on arrival), two to a dword across the lane's segments; the gradient pairs as they are.  (Until round 4
    // every Ineg had a dword of its own: 12 pieces per level at 35x35, whose landing area did not fit beside the tiles at
    // three waves per SIMD -- no hand-over for the reference's own window; 9 now.  31x31: 9 -> 7 pieces, 22 % fewer bytes.)
    // Only where it saves a piece: at 21x21 and 15x15 (one segment per lane) the template is 4 / 3 pieces either way, and the
    // conversion would cost the C2 launch 1 % for nothing (same-box A/B, profiles/r04_ab_tmpl_format.txt).
    static constexpr int NI = C::TPL * C::S;
    static constexpr bool PACKED = ((NI + 1) / 2 + C::TPL * 2 * HP + 1 + 3) / 4 < (NI + C::TPL * 2 * HP + 1 + 3) / 4;
    static constexpr int NIP = PACKED ? (NI + 1) / 2 : NI;
    static constexpr int NDW = NIP + C::TPL * 2 * HP;
    // one spare dword behind the template: lanes 0, 1, 2 carry A11, A12, A22 of the level there
    static constexpr int NQ = (NDW + 1 + 3) / 4;
    static constexpr int LEVEL = NQ * 64;              // 16-byte pieces per level
    // Waves per SIMD the kernel is built for (its launch bounds), and whether the LDS landing area of one level's
    // template (NQ KB per wave) fits beside the tiles at that occupancy: no reuse for a window where it does not
    static constexpr int WAVES = C::TPL == 1 ? 5 : 3;
    static constexpr bool FITS = (C::LDS_DW * 4 + NQ * 1024) * 4 * WAVES <= 160 * 1024;
    static constexpr int LDS_Q = FITS ? NQ * 64 : 1;
};

template <int WW, int WH>
__device__ __forceinline__ void tmpl_store(const Template<WW, WH, 1>& T, float A11, float A12, float A22,
                                           uint4* __restrict__ dst, int lane)
{
    using IO = TmplIO<WW, WH>;
    using C = Cfg<WW, WH>;
    uint32_t d[IO::NQ * 4];
#pragma unroll
    for (int i = 0; i < IO::NQ * 4; i++) d[i] = 0u;
    d[IO::NDW] = __float_as_uint(lane == 0 ? A11 : (lane == 1 ? A12 : A22));
    constexpr int kSeed = 1 << (W_BITS - 6), kShift = W_BITS - 5;
    if constexpr (IO::PACKED) {
#pragma unroll
        for (int p = 0; p < IO::NIP; p++) {
            const uint32_t lo = (uint32_t)(kSeed - T.Ineg[0][(2 * p) / C::S][(2 * p) % C::S]) >> kShift;
            uint32_t hi = 0u;
            if (2 * p + 1 < IO::NI) hi = (uint32_t)(kSeed - T.Ineg[0][(2 * p + 1) / C::S][(2 * p + 1) % C::S]) << (16 - kShift);   // iv << 16
            d[p] = lo | hi;
        }
    } else {
#pragma unroll
        for (int n = 0; n < IO::NI; n++) d[n] = (uint32_t)T.Ineg[0][n / C::S][n % C::S];
    }
#pragma unroll
    for (int k = 0; k < C::TPL; k++) {
#pragma unroll
        for (int q = 0; q < IO::HP; q++) {
            d[IO::NIP + k * 2 * IO::HP + q] = T.Ixp[0][k][q];
            d[IO::NIP + k * 2 * IO::HP + IO::HP + q] = T.Iyp[0][k][q];
        }
    }
    // non-temporal both ways (here and the fetch): 15 KB per feature that the next launch reads once must not push the
    // pyramid levels the search tiles come from out of L2
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int q = 0; q < IO::NQ; q++) {
        const u4v v = {d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]};
        __builtin_nontemporal_store(v, reinterpret_cast<u4v*>(dst + q * 64 + lane));
    }
}

template <int WW, int WH>
__device__ __forceinline__ void tmpl_unpack(Template<WW, WH, 1>& T, float& A11, float& A12, float& A22,
                                            const uint4 (&v)[TmplIO<WW, WH>::NQ])
{
    using IO = TmplIO<WW, WH>;
    using C = Cfg<WW, WH>;
    uint32_t d[IO::NQ * 4];
#pragma unroll
    for (int q = 0; q < IO::NQ; q++) { d[4 * q] = v[q].x; d[4 * q + 1] = v[q].y; d[4 * q + 2] = v[q].z; d[4 * q + 3] = v[q].w; }
    A11 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)d[IO::NDW], 0));
    A12 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)d[IO::NDW], 1));
    A22 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)d[IO::NDW], 2));
    constexpr int kSeed = 1 << (W_BITS - 6), kShift = W_BITS - 5;
#pragma unroll
    for (int n = 0; n < IO::NI; n++) {
        if constexpr (IO::PACKED) {
            const uint32_t w = d[n / 2];
            // iv << kShift out of the low / the high half of the dword
            const uint32_t sh = (n & 1) ? ((w >> (16 - kShift)) & ~((1u << kShift) - 1u)) : ((w & 0xffffu) << kShift);
            T.Ineg[0][n / C::S][n % C::S] = kSeed - (int)sh;
        } else {
            T.Ineg[0][n / C::S][n % C::S] = (int)d[n];
        }
    }
#pragma unroll
    for (int k = 0; k < C::TPL; k++) {
#pragma unroll
        for (int q = 0; q < IO::HP; q++) {
            T.Ixp[0][k][q] = d[IO::NIP + k * 2 * IO::HP + q];
            T.Iyp[0][k][q] = d[IO::NIP + k * 2 * IO::HP + IO::HP + q];
        }
    }
}

// tio: this feature's stored templates (TmplIO::LEVEL pieces per level).  tmode 2: this pass leaves the templates and
// 2x2 matrices it builds there.  tmode 1: the ones the backward pass of the pair before left there are used instead of
// building them -- fetched straight into LDS (global_load_lds_dwordx4: no registers, nothing waits), a level ahead:
// the request for level l-1 goes out when level l has been taken over into registers, and arrives while level l iterates.
// The "lk_sums" variants (LKParams::sum_mode 1 / 2: A11 .. b2 accumulated in the float lanes of OpenCV's x86 SIMD blocks,
// oracle/icelk_oracle.c lanes_a_row / lanes_b_px).  Float addition does not associate, so the order IS the result: every
// pixel's products sit in LDS in raster order (template_pixels / residual_pixels); chain c < 4 of a plane adds the products
// of the columns x = c (mod 4) below `whole`, row by row, chain 4 those of the columns from `whole` on (the scalar tail of a
// row), one float addition at a time -- a lane per chain, 5 x planes lanes busy --, and the five chains of a plane are
// folded the way the blocks fold their lanes.  The same statement as k_lk.hip's lane_chain_sums, here behind the tuned
// kernel's tiles, templates and packed pixel arithmetic: selecting a variant costs the chains, not the whole kernel.
template <int WW, int WH, int WHOLE>
__device__ __forceinline__ void chain_sums_t(const float* fsum, int planes, bool pairwise, float (&out)[3], int lane)
{
    constexpr int NPX = WW * WH;
    // a chain's terms of one window row: columns c, c + 4, ... below WHOLE (NI of them), or the NT columns from WHOLE on.
    // The terms of RB rows are fetched together (a row at a time each addition waited for its own LDS read: 8 times the
    // launch time of the default sums at 35x35), then added strictly in raster order; a slot a lane does not have holds
    // 0.0f, which leaves a sum as it is (no sum here is ever -0).
    // (terms in flight: 32 where the kernel has registers to spare, 16 for the windows with several segments per lane --
    // with 32 the 35x35 kernel spilled 525 registers and ran at half the speed of the row-at-a-time form)
    constexpr int NI = WHOLE / 4, NT = WW - WHOLE, MAXN = NI > NT ? NI : NT, INFL = Cfg<WW, WH>::TPL > 1 ? 16 : 32,
                  RB = INFL / MAXN > 0 ? INFL / MAXN : 1;
    __syncthreads();
    float acc = 0.f;
    const int p = lane / 5, c = lane - 5 * p;
    const bool busy = lane < 5 * planes;
    const int x0 = c < 4 ? c : WHOLE, stride = c < 4 ? 4 : 1, count = busy ? (c < 4 ? NI : NT) : 0;
    const float* src = fsum + (busy ? p : 0) * NPX + x0;
#pragma unroll 1
    for (int y0 = 0; y0 < WH; y0 += RB) {
        float v[RB][MAXN];
#pragma unroll
        for (int r = 0; r < RB; r++)
#pragma unroll
            for (int i = 0; i < MAXN; i++) {
                const bool ok = i < count && y0 + r < WH;
                v[r][i] = ok ? src[(y0 + r) * WW + i * stride] : 0.f;
            }
#pragma unroll
        for (int r = 0; r < RB; r++)
#pragma unroll
            for (int i = 0; i < MAXN; i++) acc = __fadd_rn(acc, v[r][i]);
    }
#pragma unroll
    for (int q = 0; q < 3; q++) {
        if (q >= planes) break;
        const float q0 = __shfl(acc, 5 * q), q1 = __shfl(acc, 5 * q + 1), q2 = __shfl(acc, 5 * q + 2), q3 = __shfl(acc, 5 * q + 3);
        const float t = __shfl(acc, 5 * q + 4);
        out[q] = pairwise ? __fadd_rn(t, __fadd_rn(__fadd_rn(q0, q2), __fadd_rn(q1, q3)))
                          : __fadd_rn(t, __fadd_rn(__fadd_rn(__fadd_rn(q0, q1), q2), q3));
    }
    __syncthreads();
}

// whole = WW & ~3 (groups of 4 pixels) or WW & ~7 (groups of 8)
template <int WW, int WH>
__device__ __forceinline__ void chain_sums(const float* fsum, int planes, int whole, bool pairwise, float (&out)[3], int lane)
{
    if (whole == (WW & ~3)) chain_sums_t<WW, WH, (WW & ~3)>(fsum, planes, pairwise, out, lane);
    else chain_sums_t<WW, WH, (WW & ~7)>(fsum, planes, pairwise, out, lane);
}

// fsum: LDS room for three planes of WW x WH float products when the launch runs under a "lk_sums" variant, else null
template <int WW, int WH>
__device__ __forceinline__ TrackResult track_point_fast(const Pyramid& PI, const Pyramid& PJ, float p0x, float p0y,
                                                        const LKParams& P, uint32_t* ldsI, uint32_t* ldsJ, int lane,
                                                        bool want_err, uint4* __restrict__ tio, int tmode, uint4* tlds,
                                                        float* fsum = nullptr)
{
    using C = Cfg<WW, WH>;
    using IO = TmplIO<WW, WH>;
    const bool reuse = IO::FITS && tmode == 1;   // the same for every lane
    int in_lds = -1;                             // level whose template is in (or on its way into) tlds
    auto fetch = [&](int level) {
        // piece q of the lane: q KB further on both sides -- the instruction's immediate offset moves the global and the
        // LDS address alike, so four pieces share one address register and one M0
        // (the level's base as a scalar pair, the lane as a 32-bit offset: no 64-bit address sits in vector registers)
        const unsigned long long sb = reinterpret_cast<unsigned long long>(tio + level * IO::LEVEL);
        const uint4* src = reinterpret_cast<const uint4*>(((unsigned long long)(unsigned)uni((int)(sb >> 32)) << 32) |
                                                          (unsigned)uni((int)sb)) + lane;
        static_for<IO::NQ>([&](auto qq) {
            constexpr int q = qq, grp = q / 4, in = q % 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + grp * 256),
                                             (__attribute__((address_space(3))) void*)(tlds + grp * 256), 16, in * 1024, 2 /* nt */);
        });
        in_lds = uni(level);
    };
    if (reuse) fetch(P.top_level);
    constexpr int S = C::S;
    constexpr int R = kMargin;
    const float half_x = (WW - 1) * 0.5f, half_y = (WH - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    // exact wave sums: the grouped form needs (pixels per lane) * (group size) * (max term) < 2^31
    constexpr bool kSmall = C::TPL * S <= 8;

    // this lane's row segments
    int trow[C::TPL], tcol[C::TPL], tlen[C::TPL], joff[C::TPL];
    uint32_t pmask[C::TPL][(S + 1) / 2];   // which of a segment's pixels belong to the window (packed gradient pairs)
#pragma unroll
    for (int k = 0; k < C::TPL; k++) {
        const int t = lane + 64 * k;
        const bool used = t < C::NTASK;   // surplus lanes work on segment 0 with every pixel masked off
        const int row = used ? t / C::NSEG : 0;
        trow[k] = row;
        tcol[k] = used ? (t - row * C::NSEG) * S : 0;
        tlen[k] = used ? (WW - tcol[k] < S ? WW - tcol[k] : S) : 0;
        joff[k] = row * C::JPD * 4 + tcol[k];
#pragma unroll
        for (int q = 0; q < (S + 1) / 2; q++)
            pmask[k][q] = (2 * q < tlen[k] ? 0xffffu : 0u) | (2 * q + 1 < tlen[k] ? 0xffff0000u : 0u);
    }

    // The window with three segments per lane (35x35) keeps ONE register per segment -- row | column << 8 | pixels
    // << 16 -- and unpack it where it is used (a few bit-field extracts per level pass / iteration): the unpacked
    // forms (4 per segment) and the pixel masks (4 per segment) were 24 registers that lived through every loop of a
    // kernel that spills at 168.  The asm statement keeps the compiler from hoisting the unpacked values back out.
    // (35x35: 168 VGPRs + 3 spilled + 102 SGPR spills + 16 B of scratch -> 158 VGPRs, 24 SGPR spills, no scratch, at the
    // same speed.  31x31 -- two segments per lane, 154 VGPRs without a spill -- does not pay for the unpacking: 1 695 us
    // per pair against 1 645, and held to 128 VGPRs for a fourth wave 1 790; profiles/r04_lk_resources.json)
    constexpr bool kPackSeg = C::TPL >= 3;
    uint32_t segp[C::TPL];
#pragma unroll
    for (int k = 0; k < C::TPL; k++) segp[k] = (uint32_t)trow[k] | ((uint32_t)tcol[k] << 8) | ((uint32_t)tlen[k] << 16);
    auto seg_rows_cols = [&](int (&row)[C::TPL], int (&col)[C::TPL], int (&len)[C::TPL]) {
#pragma unroll
        for (int k = 0; k < C::TPL; k++) {
            uint32_t v = segp[k];
            asm volatile("" : "+v"(v));
            row[k] = (int)(v & 255u);
            col[k] = (int)((v >> 8) & 255u);
            len[k] = (int)(v >> 16);
        }
    };

    TrackResult Rz;
    Rz.status = 1;
    Rz.err = 0.f;
    Rz.iters = 0;
    float sx = 0.f, sy = 0.f;  // the stored nextPts value
    Template<WW, WH, 1> T;

    for (int level = P.top_level; level >= 0; level--) {
        const Level LI = PI.lv[level];
        const Level LJ = PJ.lv[level];
        const float scale = 1.f / (float)(1 << level);
        float px = p0x * scale, py = p0y * scale;
        if (level == P.top_level) { sx = px; sy = py; }
        else { sx = sx * 2.f; sy = sy * 2.f; }
        px -= half_x; py -= half_y;
        const int ipx = uni((int)floorf(px)), ipy = uni((int)floorf(py));
        if (!origin_ok<WW, WH>(LI, ipx, ipy)) {
            if (level == 0) { Rz.status = 0; Rz.err = 0.f; }
            continue;
        }
        const Weights wi = bilinear_weights(px - (float)ipx, py - (float)ipy);

        // ---- stage the template source patch and (speculatively) the first search tile ---------------
        float nx = sx - half_x, ny = sy - half_y;
        int jx0 = 0, jy0 = 0;
        bool staged = false;
        const int ix0 = ipx - 1, iy0 = ipy - 1;
        const bool i_inside = tile_inside(LI, ix0, iy0, C::ITW, C::ITH);
        {
            const int inx = uni((int)floorf(nx)), iny = uni((int)floorf(ny));
            const bool j_ok = origin_ok<WW, WH>(LJ, inx, iny);
            const int tjx = inx - R, tjy = iny - R;
            const bool j_inside = j_ok && tile_inside(LJ, tjx, tjy, C::JTW, C::JTH);
            __syncthreads();
            TileRegs<C::IPD, C::ITH> ti;
            TileRegs<C::JPD, C::JTH> tj;
            // tiles over the frame border take the reflecting loader; either way every load of the level is in flight
            // before the first LDS write waits for one
            if (!reuse) {
                if (i_inside) tile_issue(ti, LI, ix0, iy0, lane);
                else tile_issue_reflect(ti, LI, ix0, iy0, lane);
            }
            if (j_inside) tile_issue(tj, LJ, tjx, tjy, lane);
            else if (j_ok) tile_issue_reflect(tj, LJ, tjx, tjy, lane);
            if (!reuse) tile_commit(ti, ldsI, lane);
            if (j_ok) tile_commit(tj, ldsJ, lane);
            if (j_ok) { jx0 = tjx; jy0 = tjy; staged = true; }
            __syncthreads();
        }

        // ---- template patch into registers (lk_fast_tiles.h) ------------------------------------------
        float A11, A12, A22;
        if (reuse) {
            if (in_lds != level) fetch(level);              // a level above was left before its template was needed
            __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0): the pieces have landed
            uint4 tq[IO::NQ];
#pragma unroll
            for (int q = 0; q < IO::NQ; q++) tq[q] = tlds[q * 64 + lane];
            tmpl_unpack<WW, WH>(T, A11, A12, A22, tq);
            __builtin_amdgcn_s_waitcnt(0xC07F);             // lgkmcnt(0): read out before the next level overwrites it
            if (level > 0) fetch(level - 1);
        } else {
            int a11, a12, a22;
            if constexpr (kPackSeg) {
                int row_l[C::TPL], col_l[C::TPL], len_l[C::TPL];
                uint32_t pmask_l[C::TPL][(S + 1) / 2];
                seg_rows_cols(row_l, col_l, len_l);
#pragma unroll
                for (int k = 0; k < C::TPL; k++)
#pragma unroll
                    for (int q = 0; q < (S + 1) / 2; q++)
                        pmask_l[k][q] = (2 * q < len_l[k] ? 0xffffu : 0u) | (2 * q + 1 < len_l[k] ? 0xffff0000u : 0u);
                template_pixels<WW, WH, 1, 0>(T, ldsI, (uint32_t)uni((int)pack_weights_lo(wi)), (uint32_t)uni((int)pack_weights_hi(wi)),
                                              ix0 & 3, i_inside, ipx, ipy, LI.w, LI.h, row_l, col_l, pmask_l, a11, a12, a22, fsum, len_l);
            } else
            template_pixels<WW, WH, 1, 0>(T, ldsI, (uint32_t)uni((int)pack_weights_lo(wi)), (uint32_t)uni((int)pack_weights_hi(wi)),
                                          ix0 & 3, i_inside, ipx, ipy, LI.w, LI.h, trow, tcol, pmask, a11, a12, a22, fsum, tlen);
            // |Ix*Ix| <= 4080^2 per pixel: 16-lane sums fit int32 while a lane holds <= 8 pixels
            long long s11, s12, s22;
            wave_sum3_i64<kSmall ? 16 : 1>(a11, a12, a22, s11, s12, s22);
            A11 = sum_to_float(s11) * FLT_SCALE;
            A12 = sum_to_float(s12) * FLT_SCALE;
            A22 = sum_to_float(s22) * FLT_SCALE;
            if (fsum) {
                // 3.x: groups of 4 pixels, lanes folded ((l0+l1)+l2)+l3; 4.x: groups of 8, folded (l0+l2)+(l1+l3)
                float o[3];
                chain_sums<WW, WH>(fsum, 3, P.sum_mode == 1 ? (WW & ~3) : (WW & ~7), P.sum_mode == 2, o, lane);
                A11 = __fmul_rn(o[0], FLT_SCALE); A12 = __fmul_rn(o[1], FLT_SCALE); A22 = __fmul_rn(o[2], FLT_SCALE);
            }
        }
        if (IO::FITS && tmode == 2) tmpl_store<WW, WH>(T, A11, A12, A22, tio + level * IO::LEVEL, lane);
        float D = __fsub_rn(__fmul_rn(A11, A22), __fmul_rn(A12, A12));
        const float dif = __fsub_rn(A11, A22);
        const float rad = __fadd_rn(__fmul_rn(dif, dif), __fmul_rn(__fmul_rn(4.f, A12), A12));
        const float tr = __fadd_rn(A22, A11);
        // The exact (correctly rounded sqrt and divide) minEig is only needed when it is reported or close
        // to the threshold: the hardware-approximate value differs from it by a few ulp of `tr` / (2wh).
        const float approx = (tr - __builtin_amdgcn_sqrtf(rad)) * (1.f / (float)(2 * WW * WH));
        const bool clear_pass = approx > P.min_eig_thr + tr * 2e-6f && !(P.flags & ICELK_FLAG_MIN_EIGENVALS);
        if (!uni(clear_pass)) {
            const float minEig = __fdiv_rn(__fsub_rn(tr, sqrtf(rad)), (float)(2 * WW * WH));
            if (P.flags & ICELK_FLAG_MIN_EIGENVALS) Rz.err = minEig;
            if (minEig < P.min_eig_thr) {
                if (level == 0) Rz.status = 0;
                continue;
            }
        }
        if (D < 1.1920928955078125e-07f) {
            if (level == 0) Rz.status = 0;
            continue;
        }
        D = __fdiv_rn(1.f, D);

        // ---- iterations ---------------------------------------------------------------------------
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < P.max_count; j++) {
            const int inx = uni((int)floorf(nx)), iny = uni((int)floorf(ny));
            if (!origin_ok<WW, WH>(LJ, inx, iny)) {
                if (level == 0) Rz.status = 0;
                break;
            }
            if (!staged || !tile_covers(jx0, jy0, inx, iny)) {
                jx0 = inx - R; jy0 = iny - R;
                __syncthreads();
                TileRegs<C::JPD, C::JTH> tj;
                if (tile_inside(LJ, jx0, jy0, C::JTW, C::JTH)) tile_issue(tj, LJ, jx0, jy0, lane);
                else tile_issue_reflect(tj, LJ, jx0, jy0, lane);
                tile_commit(tj, ldsJ, lane);
                __syncthreads();
                staged = true;
            }
            Rz.iters++;
            const Weights wj = bilinear_weights(nx - (float)inx, ny - (float)iny);
            const int jb = (iny - jy0) * (C::JPD * 4) + (jx0 & 3) + (inx - jx0);
            int b1, b2;
            if constexpr (kPackSeg) {
                int row_l[C::TPL], col_l[C::TPL], len_l[C::TPL], joff_l[C::TPL];
                seg_rows_cols(row_l, col_l, len_l);
#pragma unroll
                for (int k = 0; k < C::TPL; k++) joff_l[k] = row_l[k] * (C::JPD * 4) + col_l[k];
                residual_pixels<WW, WH, 1, 0, false>(T, ldsJ, jb, (uint32_t)uni((int)pack_weights_lo(wj)),
                                                     (uint32_t)uni((int)pack_weights_hi(wj)), joff_l, len_l, b1, b2, fsum, row_l, col_l);
            } else
            residual_pixels<WW, WH, 1, 0, false>(T, ldsJ, jb, (uint32_t)uni((int)pack_weights_lo(wj)),
                                                 (uint32_t)uni((int)pack_weights_hi(wj)), joff, tlen, b1, b2, fsum, trow, tcol);
            // |diff*Ix| <= 8160*4080 per pixel: 8-lane sums fit int32 while a lane holds <= 8 pixels
            long long t1, t2;
            wave_sum2_i64<kSmall ? 8 : 1>(b1, b2, t1, t2);
            float fb1 = sum_to_float(t1) * FLT_SCALE;
            float fb2 = sum_to_float(t2) * FLT_SCALE;
            if (fsum) {
                // both versions: groups of 8 pixels in 2 x 4 lanes; (q0[k] + q1[k]) pairs = chains (0 + 2) + (1 + 3)
                float o[3];
                chain_sums<WW, WH>(fsum, 2, WW & ~7, true, o, lane);
                fb1 = __fmul_rn(o[0], FLT_SCALE); fb2 = __fmul_rn(o[1], FLT_SCALE);
            }
            const float dx = __fmul_rn(__fsub_rn(__fmul_rn(A12, fb2), __fmul_rn(A22, fb1)), D);
            const float dy = __fmul_rn(__fsub_rn(__fmul_rn(A12, fb1), __fmul_rn(A11, fb2)), D);
            nx = __fadd_rn(nx, dx); ny = __fadd_rn(ny, dy);
            sx = __fadd_rn(nx, half_x); sy = __fadd_rn(ny, half_y);
            // (double)dx*dx + (double)dy*dy <= eps^2: the float value decides unless it lies in the 2^-20 band around
            // eps^2 (LKParams::eps2_lo / eps2_hi), where the exact form runs
            const float q = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
            bool conv = q < P.eps2_lo;
            if (uni(!conv && !(q > P.eps2_hi) ? 1 : 0)) {
                // a real branch (the values are wave-uniform): flattened, the six double-precision instructions of the exact
                // form ran in every iteration of every feature for a band that is 2^-19 wide
                asm volatile("" ::: "memory");
                conv = __dadd_rn(__dmul_rn((double)dx, (double)dx), __dmul_rn((double)dy, (double)dy)) <= P.eps2;
            }
            if (conv) break;
            // fabs((double)t) < 0.01 for a float t  <=>  |t| <= 0.01f, the largest float below 0.01
            if (j > 0 && fabsf(__fadd_rn(dx, pdx)) <= 0.01f && fabsf(__fadd_rn(dy, pdy)) <= 0.01f) {
                sx = __fsub_rn(sx, __fmul_rn(dx, 0.5f));
                sy = __fsub_rn(sy, __fmul_rn(dy, 0.5f));
                break;
            }
            pdx = dx; pdy = dy;
        }

        // ---- residual error at level 0 ----------------------------------------------------------------
        if (Rz.status && level == 0 && !(P.flags & ICELK_FLAG_MIN_EIGENVALS)) {
            const float qx = sx - half_x, qy = sy - half_y;
            const int iqx = uni((int)floorf(qx)), iqy = uni((int)floorf(qy));
            if (!origin_ok<WW, WH>(LJ, iqx, iqy)) {
                Rz.status = 0;
                continue;
            }
            // the mean absolute residual itself is only formed for a caller that takes it (the segment loop does not:
            // s1:323 drops `err`): one more search-tile check, 7 bilinear samples per lane and a wave sum saved
            if (!want_err) continue;
            if (!staged || !tile_covers(jx0, jy0, iqx, iqy)) {
                jx0 = iqx - R; jy0 = iqy - R;
                __syncthreads();
                TileRegs<C::JPD, C::JTH> tj;
                if (tile_inside(LJ, jx0, jy0, C::JTW, C::JTH)) tile_issue(tj, LJ, jx0, jy0, lane);
                else tile_issue_reflect(tj, LJ, jx0, jy0, lane);
                tile_commit(tj, ldsJ, lane);
                __syncthreads();
                staged = true;
            }
            const Weights we = bilinear_weights(qx - (float)iqx, qy - (float)iqy);
            const int jb = (iqy - jy0) * (C::JPD * 4) + (jx0 & 3) + (iqx - jx0);
            int es, unused;
            if constexpr (kPackSeg) {
                int row_l[C::TPL], col_l[C::TPL], len_l[C::TPL], joff_l[C::TPL];
                seg_rows_cols(row_l, col_l, len_l);
#pragma unroll
                for (int k = 0; k < C::TPL; k++) joff_l[k] = row_l[k] * (C::JPD * 4) + col_l[k];
                residual_pixels<WW, WH, 1, 0, true>(T, ldsJ, jb, (uint32_t)uni((int)pack_weights_lo(we)),
                                                    (uint32_t)uni((int)pack_weights_hi(we)), joff_l, len_l, es, unused);
            } else
            residual_pixels<WW, WH, 1, 0, true>(T, ldsJ, jb, (uint32_t)uni((int)pack_weights_lo(we)),
                                                (uint32_t)uni((int)pack_weights_hi(we)), joff, tlen, es, unused);
            const float errval = sum_to_float(sum_pick<kSmall, 8>(es));
            Rz.err = __fdiv_rn(__fmul_rn(errval, 1.f), (float)(32 * WW * WH));
        }
    }
    Rz.x = sx;
    Rz.y = sy;
    return Rz;
}

// Up to two jobs per launch.  Workgroups are dealt to the jobs in groups of 8 (one group = one workgroup per XCD, so a
// job keeps its XCD <-> contiguous-eighth mapping): groups alternate between the jobs while both have groups left, the
// longer job takes the rest.  g0, g1 = workgroups of the two jobs (multiples of 8; g1 = 0: a single job).
struct LKJobs {
    LKJob job[2];
    int g0, g1;
};

template <int WW, int WH, bool FB, bool SM = false>
__device__ __forceinline__ void lk_fast_body(const LKJobs& JJ, const LKParams& P)
{
    using C = Cfg<WW, WH>;
    __shared__ uint32_t lds[C::LDS_DW];
    __shared__ uint4 tlds[SM ? 1 : TmplIO<WW, WH>::LDS_Q];   // landing area of one level's stored template
    __shared__ float fs_[SM ? 3 * WW * WH : 1];              // "lk_sums" variants: the pixels' products, three planes
    float* const fsum = SM ? fs_ : nullptr;
    int which = 0, b = blockIdx.x;
    if (JJ.g1 > 0) {
        // The hardware deals the workgroups of a launch to the CUs of an XCD ROUND ROBIN, by count -- not to whichever CU
        // has a slot free (tools/lk_stamps_pair.py: every CU gets 76-80 of the 20 016 workgroups whatever they take).
        // Alternating the jobs group by group therefore gave all workgroups of one job to the even CUs and the other job's
        // to the odd ones (32 CUs per XCD, period 2): 128 CUs per job, and the launch lasted as long as the slower job
        // on half the chip.  Blocks of 32 groups (one workgroup per CU of every XCD) alternate instead, so that every CU
        // works through both jobs; what is left of the shorter job after its whole blocks alternates group by group.
        const int G = b >> 3, m = (JJ.g0 < JJ.g1 ? JJ.g0 : JJ.g1) >> 3;   // groups interleaved per job
        const int mb = m & ~31;
        int Gj;
        if (G < 2 * mb) {
            which = (G >> 5) & 1;
            Gj = ((G >> 6) << 5) | (G & 31);
        } else if (G < 2 * m) {
            const int r = G - 2 * mb;
            which = r & 1;
            Gj = mb + (r >> 1);
        } else {
            which = JJ.g0 < JJ.g1 ? 1 : 0;
            Gj = G - m;
        }
        b = (Gj << 3) | (b & 7);
    }
    const LKJob& J = JJ.job[which];
    const LKBuffers& B = J.B;
    // the same for every lane (a table entry fetched by a vector load): keep it in a scalar register
    const int f = __builtin_amdgcn_readfirstlane(launch_slot(B, b, B.n_dev ? *B.n_dev : J.n));
    if (f < 0) return;
    if (B.seg_alive && !B.seg_alive[f]) return;
    const int lane = threadIdx.x;
    uint32_t* ldsI = lds;
    uint32_t* ldsJ = lds + C::I_DW;
    const float p0x = B.p_in[2 * f], p0y = B.p_in[2 * f + 1];
    if (lane == 0) stamp(B, 0);
    using IO = TmplIO<WW, WH>;
    const size_t t_off = (size_t)f * ((size_t)B.tmpl_levels * IO::LEVEL);   // pieces before this track's
    const TrackResult r1 = track_point_fast<WW, WH>(J.I, J.J, p0x, p0y, P, ldsI, ldsJ, lane, B.err_fwd != nullptr,
                                                    const_cast<uint4*>(static_cast<const uint4*>(B.tmpl_in)) + t_off,
                                                    !SM && B.tmpl_in ? 1 : 0, tlds, fsum);
    if (lane == 0) {
        if (B.p_fwd) { B.p_fwd[2 * f] = r1.x; B.p_fwd[2 * f + 1] = r1.y; }
        if (B.st_fwd) B.st_fwd[f] = (uint8_t)r1.status;
        if (B.err_fwd) B.err_fwd[f] = r1.err;
        if (B.iters && !FB) B.iters[f] = (uint32_t)r1.iters;
    }
    if (FB) {
        const TrackResult r2 = track_point_fast<WW, WH>(J.J, J.I, r1.x, r1.y, P, ldsI, ldsJ, lane, B.err_bwd != nullptr,
                                                        static_cast<uint4*>(B.tmpl_out) + t_off, !SM && B.tmpl_out ? 2 : 0, tlds,
                                                        fsum);
        if (lane == 0) {
            if (B.p_bwd) { B.p_bwd[2 * f] = r2.x; B.p_bwd[2 * f + 1] = r2.y; }
            if (B.st_bwd) B.st_bwd[f] = (uint8_t)r2.status;
            if (B.err_bwd) B.err_bwd[f] = r2.err;
            if (B.iters) B.iters[f] = (uint32_t)r1.iters | ((uint32_t)r2.iters << 16);
            const float d = fb_distance(p0x, p0y, r2.x, r2.y, P.dist_form);
            if (B.dist) B.dist[f] = d;
            if (B.valid) B.valid[f] = d < P.fb_thr ? 1 : 0;
            if (B.seg_alive) seg_append(B, f, r1.x, r1.y, d, d < P.fb_thr);
        }
    }
    if (lane == 0) stamp(B, 1);
}

// Waves per SIMD the register allocation aims at: 21x21 needs 86 VGPRs (5 waves fit), 31x31 144 (3), 35x35 180 -- held to
// 168 for a third wave at the price of 16 B of scratch: REF 654 -> 770 pairs/s.  Forcing more (21x21 at 8, 31x31 at 4) spills
// into the loops and loses.
template <int WW, int WH, bool FB>
__global__ __launch_bounds__(64, 3) void k_lk_fast(LKJobs JJ, LKParams P)
{
    lk_fast_body<WW, WH, FB>(JJ, P);
}

// One row segment per lane (21x21, 15x15): held to 88 VGPRs.  Five waves of 88 leave 72 registers of a SIMD free, and the
// kernels that run beside a tracker launch (one-wave pyramid: 70, the min-distance chain) fit into those without waiting
// for a tracker wave to retire; with the template reuse the allocator would otherwise take 93 (still five waves, but 32
// left over: the reuse then loses more than it wins).  amdgpu_num_vgpr counts half of the unified VGPR+AGPR file.
template <int WW, int WH, bool FB>
__global__ __launch_bounds__(64, 4) __attribute__((amdgpu_num_vgpr(44))) void k_lk_fast88(LKJobs JJ, LKParams P)
{
    lk_fast_body<WW, WH, FB>(JJ, P);
}

// the same kernels under a "lk_sums" variant: the float-lane chains (chain_sums) and LDS for the pixels' products
template <int WW, int WH, bool FB>
__global__ __launch_bounds__(64, 2) void k_lk_fast_sums(LKJobs JJ, LKParams P)
{
    lk_fast_body<WW, WH, FB, true>(JJ, P);
}

int grid_of(const LKBuffers& B, int n) { return B.order ? (n + 15) & ~7 : (n + 7) & ~7; }

template <int WW, int WH>
void launch_fast(hipStream_t s, const LKJob& a, const LKJob* b, const LKParams& P, bool fb)
{
    LKJobs JJ;
    JJ.job[0] = a;
    JJ.job[1] = b ? *b : a;
    JJ.g0 = grid_of(a.B, a.n);
    JJ.g1 = b ? grid_of(b->B, b->n) : 0;
    const int grid = JJ.g0 + JJ.g1;
    if (P.sum_mode) {
        if (fb) hipLaunchKernelGGL((k_lk_fast_sums<WW, WH, true>), dim3(grid), dim3(64), 0, s, JJ, P);
        else hipLaunchKernelGGL((k_lk_fast_sums<WW, WH, false>), dim3(grid), dim3(64), 0, s, JJ, P);
        return;
    }
    if constexpr (Cfg<WW, WH>::TPL == 1) {
        if (fb) hipLaunchKernelGGL((k_lk_fast88<WW, WH, true>), dim3(grid), dim3(64), 0, s, JJ, P);
        else hipLaunchKernelGGL((k_lk_fast88<WW, WH, false>), dim3(grid), dim3(64), 0, s, JJ, P);
    } else {
        if (fb) hipLaunchKernelGGL((k_lk_fast<WW, WH, true>), dim3(grid), dim3(64), 0, s, JJ, P);
        else hipLaunchKernelGGL((k_lk_fast<WW, WH, false>), dim3(grid), dim3(64), 0, s, JJ, P);
    }
}

bool dispatch_fast(hipStream_t s, const LKJob& a, const LKJob* b, const LKParams& P, bool fb)
{
    if (P.flags & ICELK_FLAG_INITIAL_FLOW) return false;
    if (P.win_w == 21 && P.win_h == 21) launch_fast<21, 21>(s, a, b, P, fb);
    else if (P.win_w == 31 && P.win_h == 31) launch_fast<31, 31>(s, a, b, P, fb);
    else if (P.win_w == 35 && P.win_h == 35) launch_fast<35, 35>(s, a, b, P, fb);
    else if (P.win_w == 15 && P.win_h == 15) launch_fast<15, 15>(s, a, b, P, fb);
    else return false;
    return true;
}

}  // namespace

int lk_template_quads(int win_w, int win_h)
{
    if (win_w == 21 && win_h == 21) return TmplIO<21, 21>::FITS ? TmplIO<21, 21>::NQ : 0;
    if (win_w == 31 && win_h == 31) return TmplIO<31, 31>::FITS ? TmplIO<31, 31>::NQ : 0;
    if (win_w == 35 && win_h == 35) return TmplIO<35, 35>::FITS ? TmplIO<35, 35>::NQ : 0;
    if (win_w == 15 && win_h == 15) return TmplIO<15, 15>::FITS ? TmplIO<15, 15>::NQ : 0;
    return 0;
}

bool lk_fast_eligible(const LKParams& P)
{
    if (P.flags & (ICELK_FLAG_INITIAL_FLOW | ICELK_FLAG_GENERIC_KERNEL | ICELK_FLAG_MULTI_PER_WAVE)) return false;
    return P.sum_mode == 0 && lk_template_quads(P.win_w, P.win_h) > 0;
}

// Returns true when a specialised kernel exists for this window (and INITIAL_FLOW is not requested).
bool launch_lk_fast(hipStream_t s, const Pyramid& I, const Pyramid& J, const LKBuffers& B, int n, const LKParams& P,
                    bool fb)
{
    LKJob a;
    a.I = I;
    a.J = J;
    a.B = B;
    a.n = n;
    return dispatch_fast(s, a, nullptr, P, fb);
}

bool launch_lk_pair(hipStream_t s, const LKJob& a, const LKJob& b, const LKParams& P)
{
    if (P.flags & (ICELK_FLAG_GENERIC_KERNEL | ICELK_FLAG_MULTI_PER_WAVE)) return false;
    if (a.n <= 0 || b.n <= 0) return false;
    return dispatch_fast(s, a, &b, P, true);
}

}  // namespace icelk
