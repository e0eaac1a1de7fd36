// lk_fast_tiles.h -- building blocks shared by the window-size-specialised Lucas-Kanade kernels
// (k_lk_fast.hip: one feature per wave; k_lk_multi.hip: several features per wave): compile-time window
// geometry, staging of image tiles in LDS as aligned dwords, byte -> 16-bit-pair widening for v_dot2_i32_i16.
#pragma once
#include <type_traits>

#include "lk_common.h"

namespace icelk {
namespace lkf {

using namespace lk;

// Search-tile margin R: the estimate may move +-R px from where the tile was staged before it is staged again.  Results
// do not depend on it.  Measured (C2 / REF / C5 pairs/s): R = 6: 5 040 / 611 / 470; 3: 5 300 / 640 / 530; 2: 5 280 / 652 /
// 547; 1: 5 360 / 654 / 552; 0: 5 240 / 645 / 534.  A level's first guess comes from the level above and lands within a
// pixel of the answer, so a wide tile is loaded (and its addresses computed, and its registers held) for nothing; with
// R = 1 the 21x21 search tile is 24x24 B = 3 rounds of loads instead of 6 (34x34).
constexpr int kMargin = kLkTileMargin;

constexpr int pick_seg(int ww, int wh)
{
    int best = 8, best_cost = 1 << 30;
    for (int s = 5; s <= 8; s++) {
        const int nseg = (ww + s - 1) / s;
        const int tpl = (nseg * wh + 63) / 64;
        const int cost = tpl * (s + 3);
        if (cost < best_cost) { best_cost = cost; best = s; }
    }
    return best;
}

template <int WW, int WH>
struct Cfg {
    static constexpr int S = pick_seg(WW, WH);       // pixels per row segment
    static constexpr int NSEG = (WW + S - 1) / S;    // segments per window row
    static constexpr int NTASK = NSEG * WH;
    static constexpr int TPL = (NTASK + 63) / 64;    // segments per lane
    static constexpr int ITW = WW + 3, ITH = WH + 3;  // template source patch (1-px ring for Scharr + bilinear)
    static constexpr int IPD = (ITW + 2) / 4 + 1;     // LDS row pitch in dwords (any 4-byte phase)
    static constexpr int JTW = WW + 1 + 2 * kMargin, JTH = WH + 1 + 2 * kMargin;
    static constexpr int JPD = (JTW + 2) / 4 + 1;
    static constexpr int I_DW = IPD * ITH, J_DW = JPD * JTH;
    static constexpr int LDS_DW = I_DW + J_DW + 8;    // +8: realignment reads may run 3 dwords past a row
};

// ---- tile staging ---------------------------------------------------------------------------------
// A tile whose top-left image pixel is (x0, y0) is kept in LDS as the aligned dwords that cover each of
// its rows: LDS byte (r*PD*4 + (x0 & 3) + tx) holds image pixel (x0 + tx, y0 + r).
template <int PD, int TH>
struct TileRegs {
    static constexpr int N = (PD * TH + 63) / 64;
    uint32_t v[N];
};

template <int PD, int TH>
__device__ __forceinline__ void tile_issue(TileRegs<PD, TH>& t, const Level& L, int x0, int y0, int lane)
{
    // uniform 64-bit base + per-lane 32-bit offset (r * pitch + 4c < 2^24): one 24-bit mad per load, and no
    // predication -- surplus lanes of the last round re-load the tile's last dword
    const uint8_t* base = L.ptr + (size_t)y0 * L.pitch + (x0 & ~3);
#pragma unroll
    for (int m = 0; m < TileRegs<PD, TH>::N; m++) {
        int i = lane + 64 * m;
        i = i < PD * TH ? i : PD * TH - 1;
        const int r = i / PD, c = i - r * PD;
        const unsigned off = (unsigned)__mul24(r, L.pitch) + 4u * (unsigned)c;
        t.v[m] = *reinterpret_cast<const uint32_t*>(base + off);
    }
}

template <int PD, int TH>
__device__ __forceinline__ void tile_commit(const TileRegs<PD, TH>& t, uint32_t* lds, int lane)
{
#pragma unroll
    for (int m = 0; m < TileRegs<PD, TH>::N; m++) {
        int i = lane + 64 * m;
        i = i < PD * TH ? i : PD * TH - 1;   // same value written twice: harmless
        lds[i] = t.v[m];
    }
}

// Tiles that reach over the image border (features within a window of the frame edge at this level): the same
// aligned dwords, fetched with reflection (BORDER_REFLECT_101).  Rows reflect as a whole; a dword whose four columns
// lie inside the row is one load as usual; the few dwords that straddle the left or right edge are put together
// from four reflected byte loads.  Two rounds of loads in flight per tile, whatever its size -- the byte-wise loop
// this replaces paid the memory latency TW*TH/64 times in a row and made the border features (9 % of them at
// 4000x3000, 21x21, 4 levels) the stragglers of every launch: 92 us against 50 us for an interior feature.
template <int PD, int TH>
__device__ __forceinline__ void tile_issue_reflect(TileRegs<PD, TH>& t, const Level& L, int x0, int y0, int lane)
{
    constexpr int N = TileRegs<PD, TH>::N;
    const int xa = x0 & ~3;
    const uint8_t* row[N];
    int gx[N];
    bool edge[N];
#pragma unroll
    for (int m = 0; m < N; m++) {
        int i = lane + 64 * m;
        i = i < PD * TH ? i : PD * TH - 1;
        const int r = i / PD, c = i - r * PD;
        row[m] = L.ptr + (size_t)reflect101(y0 + r, L.h) * L.pitch;
        gx[m] = xa + 4 * c;
        edge[m] = !(L.w >= 4 && (unsigned)gx[m] <= (unsigned)(L.w - 4));
        t.v[m] = edge[m] ? 0u : *reinterpret_cast<const uint32_t*>(row[m] + gx[m]);
    }
    bool any = false;
#pragma unroll
    for (int m = 0; m < N; m++) any |= edge[m];
    if (__builtin_amdgcn_ballot_w64(any) == 0) return;
    uint8_t q[N][4];
#pragma unroll
    for (int m = 0; m < N; m++)
#pragma unroll
        for (int k = 0; k < 4; k++) q[m][k] = edge[m] ? row[m][reflect101(gx[m] + k, L.w)] : (uint8_t)0;
#pragma unroll
    for (int m = 0; m < N; m++)
        if (edge[m]) t.v[m] = (uint32_t)q[m][0] | ((uint32_t)q[m][1] << 8) | ((uint32_t)q[m][2] << 16) | ((uint32_t)q[m][3] << 24);
}

template <int PD, int TW, int TH>
__device__ __forceinline__ void tile_border(uint32_t* lds, const Level& L, int x0, int y0, int lane)
{
    TileRegs<PD, TH> t;
    tile_issue_reflect(t, L, x0, y0, lane);
    tile_commit(t, lds, lane);
}

__device__ __forceinline__ bool tile_inside(const Level& L, int x0, int y0, int tw, int th)
{
    // 0 <= x0 <= w - tw and 0 <= y0 <= h - th, as two unsigned compares (a level smaller than the tile fails)
    return L.w >= tw && L.h >= th && (unsigned)x0 <= (unsigned)(L.w - tw) && (unsigned)y0 <= (unsigned)(L.h - th);
}

// OpenCV's bounds test of a window origin: !(x < -W || x >= cols || y < -H || y >= rows)
template <int WW, int WH>
__device__ __forceinline__ bool origin_ok(const Level& L, int x, int y)
{
    return (unsigned)(x + WW) < (unsigned)(L.w + WW) && (unsigned)(y + WH) < (unsigned)(L.h + WH);
}

// the staged search tile (origin jx0, jy0, margin R on every side) still covers a window at (x, y)
__device__ __forceinline__ bool tile_covers(int jx0, int jy0, int x, int y)
{
    return (unsigned)(x - jx0) <= 2u * kMargin && (unsigned)(y - jy0) <= 2u * kMargin;
}

typedef short v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2s as_v2s(uint32_t x) { return __builtin_bit_cast(v2s, x); }
__device__ __forceinline__ uint32_t as_u32(v2s x) { return __builtin_bit_cast(uint32_t, x); }

// NX stream-aligned dwords (bytes 0 .. 4*NX-1) starting at byte offset `off` of an LDS row
template <int NX>
__device__ __forceinline__ void row_dwords(const uint32_t* row, int off, uint32_t (&X)[NX])
{
    const uint32_t* p = row + (off >> 2);
    const int sh = off & 3;
    uint32_t d[NX + 1];
#pragma unroll
    for (int i = 0; i <= NX; i++) d[i] = p[i];
#pragma unroll
    for (int i = 0; i < NX; i++) X[i] = __builtin_amdgcn_alignbyte(d[i + 1], d[i], sh);
}

// bytes (M, M+1) of the stream widened to a pair of 16-bit lanes: (byte M) | (byte M+1) << 16
template <int M, int NX>
__device__ __forceinline__ v2s byte_pair(const uint32_t (&X)[NX])
{
    constexpr int i = M / 4, r = M % 4;
    static_assert(i < NX && (r < 3 || i + 1 < NX), "pair outside the loaded dwords");
    if constexpr (r == 0) return as_v2s(__builtin_amdgcn_perm(0u, X[i], 0x0c010c00u));
    else if constexpr (r == 1) return as_v2s(__builtin_amdgcn_perm(0u, X[i], 0x0c020c01u));
    else if constexpr (r == 2) return as_v2s(__builtin_amdgcn_perm(0u, X[i], 0x0c030c02u));
    else return as_v2s(__builtin_amdgcn_perm(X[i + 1 < NX ? i + 1 : i], X[i], 0x0c040c03u));
}

// (a.y, b.x): the pair one 16-bit lane further along
__device__ __forceinline__ v2s pair_shift(v2s a, v2s b)
{
    return as_v2s(__builtin_amdgcn_alignbit(as_u32(b), as_u32(a), 16));
}

__device__ __forceinline__ int dot2(v2s a, v2s b, int c) { return __builtin_amdgcn_sdot2(a, b, c, false); }

template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

// packed weight pairs (w00 | w01 << 16), (w10 | w11 << 16) for v_dot2_i32_i16
__device__ __forceinline__ uint32_t pack_weights_lo(const Weights& w) { return ((uint32_t)w.w00 & 0xffffu) | ((uint32_t)w.w01 << 16); }
__device__ __forceinline__ uint32_t pack_weights_hi(const Weights& w) { return ((uint32_t)w.w10 & 0xffffu) | ((uint32_t)w.w11 << 16); }

// ---- pixel work shared by the one-feature-per-wave and the several-features-per-wave kernels ---------------------------
// The template of a feature: I as a dot2 accumulator seed, Ix / Iy as packed 16-bit pairs of neighbouring pixels.
template <int WW, int WH, int F>
struct Template {
    using C = Cfg<WW, WH>;
    int Ineg[F][C::TPL][C::S];                       // 256 - (I << 9): dot2 accumulator seed of the residual
    uint32_t Ixp[F][C::TPL][(C::S + 1) / 2];         // (Ix[2q], Ix[2q+1]) as 16-bit pairs
    uint32_t Iyp[F][C::TPL][(C::S + 1) / 2];
};

__device__ __forceinline__ uint32_t pack16(int lo, int hi)
{
    return __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x05040100u);
}

// v_dot2_i32_i16 in its three-address (VOP3P) form.  The compiler only ever selects the two-address v_dot2c, which
// overwrites its accumulator: with a seed that is needed again (the template value of the residual, a rounding
// constant) that costs a v_mov per dot product.  b and c may be scalars (weights, constants): at most one of them is.
__device__ __forceinline__ int dot2_vsv(v2s a, uint32_t b_scalar, int c)
{
    int d;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(as_u32(a)), "s"(b_scalar), "v"(c));
    return d;
}
__device__ __forceinline__ int dot2_vvs(v2s a, uint32_t b, int c_scalar)
{
    int d;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(as_u32(a)), "v"(b), "s"(c_scalar));
    return d;
}

// ---- pixel phase: template of ONE feature -------------------------------------------------------------------
// Everything that is the same for all lanes arrives as a scalar (readlane of the owning block's registers).
// fsum != nullptr (the "lk_sums" variants, LKParams::sum_mode): every window pixel's three products also go to LDS as
// floats, in raster order, for the lane-ordered float sums of chain_sums (k_lk_fast.hip); tlen = pixels of each segment
// that lie inside the window.
template <int WW, int WH, int F, int FI>
__device__ __forceinline__ void template_pixels(Template<WW, WH, F>& T, const uint32_t* ldsI, uint32_t W0u, uint32_t W1u, int ics,
                                                bool i_inside, int ipx, int ipy, int liw, int lih,
                                                const int (&trow)[Cfg<WW, WH>::TPL], const int (&tcol)[Cfg<WW, WH>::TPL],
                                                const uint32_t (&pmask)[Cfg<WW, WH>::TPL][(Cfg<WW, WH>::S + 1) / 2], int& a11,
                                                int& a12, int& a22, float* fsum = nullptr, const int* tlen = nullptr)
{
    using C = Cfg<WW, WH>;
    constexpr int S = C::S;
    // the second weight pair as a VGPR (a three-address dot2 takes one scalar), the rounding constants as scalars
    const uint32_t W1v = W1u;
    const int kRoundI = 1 << (W_BITS - 6), kRoundD = 1 << (W_BITS - 1);
    a11 = 0; a12 = 0; a22 = 0;
#pragma unroll
    for (int k = 0; k < C::TPL; k++) {
        constexpr int NP = (S + 4) / 2;   // even-aligned column pairs covering columns 0 .. S+2
        constexpr int ND = (S + 2) / 2;   // derivative pairs covering derivative columns 0 .. S
        v2s E[4][NP];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            uint32_t X[3];
            row_dwords<3>(ldsI + (trow[k] + r) * C::IPD, ics + tcol[k], X);
            static_for<NP>([&](auto kk) { E[r][kk] = byte_pair<2 * kk, 3>(X); });
        }
        // I samples first (they need source rows 1 and 2 only): columns (j+1, j+2) are the pair E[.][(j+1)/2] for odd j
        // and the pair one 16-bit lane further along for even j
        static_for<S>([&](auto jj) {
            constexpr int j = jj;
            v2s s1, s2;
            if constexpr (j % 2 == 1) {
                s1 = E[1][(j + 1) / 2]; s2 = E[2][(j + 1) / 2];
            } else {
                s1 = pair_shift(E[1][j / 2], E[1][j / 2 + 1]); s2 = pair_shift(E[2][j / 2], E[2][j / 2 + 1]);
            }
            const int iv = dot2(s2, as_v2s(W1v), dot2_vvs(s1, W0u, kRoundI)) >> (W_BITS - 5);
            T.Ineg[FI][k][j] = (1 << (W_BITS - 6)) - (iv << (W_BITS - 5));
        });
        // Scharr, two columns per instruction: t0 = 3*(a + c) + 10*b, t1 = c - a down the rows, then
        // dx = t0[i+2] - t0[i], dy = 3*(t1[i+2] + t1[i]) + 10*t1[i+1] along the row
        v2s dxp[2][ND], dyp[2][ND];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            v2s t0[NP], t1[NP];
#pragma unroll
            for (int q = 0; q < NP; q++) {
                t0[q] = (E[r][q] + E[r + 2][q]) * (short)3 + E[r + 1][q] * (short)10;
                t1[q] = E[r + 2][q] - E[r][q];
            }
#pragma unroll
            for (int d = 0; d < ND; d++) {
                dxp[r][d] = t0[d + 1] - t0[d];
                dyp[r][d] = (t1[d + 1] + t1[d]) * (short)3 + pair_shift(t1[d], t1[d + 1]) * (short)10;
            }
            if (!i_inside) {
                // derivative image is zero outside the frame (BORDER_CONSTANT), SURVEY.md A.4
                const int gy = ipy + trow[k] + r;
                const bool row_in = gy >= 0 && gy < lih;
#pragma unroll
                for (int d = 0; d < ND; d++) {
                    const int gx = ipx + tcol[k] + 2 * d;
                    const bool in0 = row_in && gx >= 0 && gx < liw, in1 = row_in && gx + 1 >= 0 && gx + 1 < liw;
                    const uint32_t m = (in0 ? 0xffffu : 0u) | (in1 ? 0xffff0000u : 0u);
                    dxp[r][d] = as_v2s(as_u32(dxp[r][d]) & m);
                    dyp[r][d] = as_v2s(as_u32(dyp[r][d]) & m);
                }
            }
        }
        int ixs[S], iys[S];
        static_for<S>([&](auto jj) {
            constexpr int j = jj;
            // derivative samples: derivative columns (j, j+1) of derivative rows 0 and 1
            v2s gx0, gx1, gy0, gy1;
            if constexpr (j % 2 == 0) {
                gx0 = dxp[0][j / 2]; gx1 = dxp[1][j / 2]; gy0 = dyp[0][j / 2]; gy1 = dyp[1][j / 2];
            } else {
                gx0 = pair_shift(dxp[0][j / 2], dxp[0][j / 2 + 1]); gx1 = pair_shift(dxp[1][j / 2], dxp[1][j / 2 + 1]);
                gy0 = pair_shift(dyp[0][j / 2], dyp[0][j / 2 + 1]); gy1 = pair_shift(dyp[1][j / 2], dyp[1][j / 2 + 1]);
            }
            ixs[j] = dot2(gx1, as_v2s(W1v), dot2_vvs(gx0, W0u, kRoundD)) >> W_BITS;
            iys[j] = dot2(gy1, as_v2s(W1v), dot2_vvs(gy0, W0u, kRoundD)) >> W_BITS;
        });
        if (fsum) {
            constexpr int NPX = WW * WH;
            static_for<S>([&](auto jj) {
                constexpr int j = jj;
                if (j < tlen[k]) {
                    const int idx = trow[k] * WW + tcol[k] + j;
                    fsum[idx] = (float)(ixs[j] * ixs[j]);
                    fsum[NPX + idx] = (float)(ixs[j] * iys[j]);
                    fsum[2 * NPX + idx] = (float)(iys[j] * iys[j]);
                }
            });
        }
        // gradient pairs for the dot2 form of the residual sums (pixels beyond the window's right edge and the
        // surplus lanes' pixels are masked off here, once, with the lane's constant masks), and the 2x2 matrix sums on
        // the same pairs
        static_for<(S + 1) / 2>([&](auto qq) {
            constexpr int q = qq;
            uint32_t px2, py2;
            if constexpr (2 * q + 1 < S) {
                px2 = pack16(ixs[2 * q], ixs[2 * q + 1]);
                py2 = pack16(iys[2 * q], iys[2 * q + 1]);
            } else {
                px2 = (uint32_t)ixs[2 * q];
                py2 = (uint32_t)iys[2 * q];
            }
            px2 &= pmask[k][q];
            py2 &= pmask[k][q];
            T.Ixp[FI][k][q] = px2;
            T.Iyp[FI][k][q] = py2;
            a11 = dot2(as_v2s(px2), as_v2s(px2), a11);
            a12 = dot2(as_v2s(px2), as_v2s(py2), a12);
            a22 = dot2(as_v2s(py2), as_v2s(py2), a22);
        });
    }
}

// ---- pixel phase: residual of ONE feature at the window origin encoded in `jb` ------------------------------------
// jb = byte offset of the window's first pixel inside the feature's staged search tile (wave-uniform).
// ERR = false: b1 = sum diff*Ix, b2 = sum diff*Iy.   ERR = true: b1 = sum |diff| over the real window pixels.
// fsum != nullptr: the two products of every window pixel also go to LDS as floats (int32 -> float, as _mm_cvtepi32_ps),
// in raster order (trow, tcol = the segments' window rows / first columns).
template <int WW, int WH, int F, int FI, bool ERR>
__device__ __forceinline__ void residual_pixels(const Template<WW, WH, F>& T, const uint32_t* ldsJ, int jb, uint32_t V0u,
                                                uint32_t V1u, const int (&joff)[Cfg<WW, WH>::TPL],
                                                const int (&tlen)[Cfg<WW, WH>::TPL], int& b1, int& b2, float* fsum = nullptr,
                                                const int* trow = nullptr, const int* tcol = nullptr)
{
    using C = Cfg<WW, WH>;
    constexpr int S = C::S;
    const uint32_t V1v = V1u;   // one weight pair rides as a scalar, the other in a VGPR
    b1 = 0; b2 = 0;
#pragma unroll
    for (int k = 0; k < C::TPL; k++) {
        constexpr int NXJ = (S + 1 + 3) / 4;
        uint32_t Y0[NXJ], Y1[NXJ];
        const int off = jb + joff[k];
        row_dwords<NXJ>(ldsJ, off, Y0);
        row_dwords<NXJ>(ldsJ + C::JPD, off, Y1);
        int diff[S];
        static_for<S>([&](auto qq) {
            constexpr int q = qq;
            // ((J bilinear + 256) >> 9) - I, with 256 - (I << 9) as the accumulator seed
            diff[q] = dot2(byte_pair<q, NXJ>(Y1), as_v2s(V1v), dot2_vsv(byte_pair<q, NXJ>(Y0), V0u, T.Ineg[FI][k][q])) >>
                      (W_BITS - 5);
        });
        if constexpr (ERR) {
            static_for<S>([&](auto qq) {
                constexpr int q = qq;
                b1 += q < tlen[k] ? (diff[q] < 0 ? -diff[q] : diff[q]) : 0;
            });
        } else {
            if (fsum) {
                constexpr int NPX = WW * WH;
                static_for<S>([&](auto qq) {
                    constexpr int q = qq;
                    if (q < tlen[k]) {
                        const uint32_t gx = T.Ixp[FI][k][q / 2], gy = T.Iyp[FI][k][q / 2];
                        const int ix = (q & 1) ? ((int)gx >> 16) : (int)(short)(gx & 0xffffu);
                        const int iy = (q & 1) ? ((int)gy >> 16) : (int)(short)(gy & 0xffffu);
                        const int idx = trow[k] * WW + tcol[k] + q;
                        fsum[idx] = (float)(diff[q] * ix);
                        fsum[NPX + idx] = (float)(diff[q] * iy);
                    }
                });
            }
            static_for<(S + 1) / 2>([&](auto qq) {
                constexpr int q = qq;
                uint32_t dp;
                if constexpr (2 * q + 1 < S) dp = pack16(diff[2 * q], diff[2 * q + 1]);
                else dp = (uint32_t)diff[2 * q];   // the partner half of the gradient pair is zero
                b1 = dot2(as_v2s(dp), as_v2s(T.Ixp[FI][k][q]), b1);
                b2 = dot2(as_v2s(dp), as_v2s(T.Iyp[FI][k][q]), b2);
            });
        }
    }
}

}  // namespace lkf
}  // namespace icelk
