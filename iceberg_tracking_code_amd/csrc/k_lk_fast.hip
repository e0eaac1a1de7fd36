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
#include <type_traits>

#include "lk_common.h"

namespace icelk {

namespace {

using namespace lk;

constexpr int kMargin = 6;  // search-tile margin R: the estimate may move +-R px before a restage

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

// border tiles: byte-wise with reflect-101 (only features within a window of the image edge)
template <int PD, int TW, int TH>
__device__ __forceinline__ void tile_border(uint32_t* lds, const Level& L, int x0, int y0, int lane)
{
    uint8_t* b = reinterpret_cast<uint8_t*>(lds);
    const int cs = x0 & 3;
    for (int i = lane; i < TW * TH; i += 64) {
        const int ty = i / TW, tx = i - ty * TW;
        b[ty * PD * 4 + cs + tx] = L.ptr[(size_t)reflect101(y0 + ty, L.h) * L.pitch + reflect101(x0 + tx, L.w)];
    }
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

template <int WW, int WH>
__device__ __forceinline__ TrackResult track_point_fast(const Pyramid& PI, const Pyramid& PJ, float p0x, float p0y,
                                                        const LKParams& P, uint32_t* ldsI, uint32_t* ldsJ, int lane)
{
    using C = Cfg<WW, WH>;
    constexpr int S = C::S;
    constexpr int R = kMargin;
    const float half_x = (WW - 1) * 0.5f, half_y = (WH - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    // exact wave sums: the grouped form needs (pixels per lane) * (group size) * (max term) < 2^31
    constexpr bool kSmall = C::TPL * S <= 8;

    // this lane's row segments
    int trow[C::TPL], tcol[C::TPL], tlen[C::TPL];
#pragma unroll
    for (int k = 0; k < C::TPL; k++) {
        const int t = lane + 64 * k;
        const bool used = t < C::NTASK;   // surplus lanes work on segment 0 with every pixel masked off
        const int row = used ? t / C::NSEG : 0;
        trow[k] = row;
        tcol[k] = used ? (t - row * C::NSEG) * S : 0;
        tlen[k] = used ? (WW - tcol[k] < S ? WW - tcol[k] : S) : 0;
    }

    TrackResult Rz;
    Rz.status = 1;
    Rz.err = 0.f;
    float sx = 0.f, sy = 0.f;  // the stored nextPts value

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
            if (i_inside) tile_issue(ti, LI, ix0, iy0, lane);
            if (j_inside) tile_issue(tj, LJ, tjx, tjy, lane);
            if (i_inside) tile_commit(ti, ldsI, lane);
            else tile_border<C::IPD, C::ITW, C::ITH>(ldsI, LI, ix0, iy0, lane);
            if (j_inside) tile_commit(tj, ldsJ, lane);
            else if (j_ok) tile_border<C::JPD, C::JTW, C::JTH>(ldsJ, LJ, tjx, tjy, lane);
            if (j_ok) { jx0 = tjx; jy0 = tjy; staged = true; }
            __syncthreads();
        }

        // ---- template patch into registers ---------------------------------------------------------
        //   Ineg = 256 - (I << 9): the template value pre-loaded into the dot2 accumulator of the residual
        //   Ixv, Iyv = bilinear Scharr derivatives (int16 range)
        const v2s W0 = {(short)wi.w00, (short)wi.w01}, W1 = {(short)wi.w10, (short)wi.w11};
        int Ineg[C::TPL][S], Ixv[C::TPL][S], Iyv[C::TPL][S];
        int a11 = 0, a12 = 0, a22 = 0;
        const int ics = ix0 & 3;
#pragma unroll
        for (int k = 0; k < C::TPL; k++) {
            constexpr int NP = (S + 4) / 2;   // even-aligned column pairs covering columns 0 .. S+2
            constexpr int ND = (S + 2) / 2;   // derivative pairs covering derivative columns 0 .. S
            v2s E[4][NP];
            uint32_t X1[3], X2[3];            // aligned dwords of source rows 1 and 2 (for the I samples)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                uint32_t X[3];
                row_dwords<3>(ldsI + (trow[k] + r) * C::IPD, ics + tcol[k], X);
                static_for<NP>([&](auto kk) { E[r][kk] = byte_pair<2 * kk, 3>(X); });
                if (r == 1) { X1[0] = X[0]; X1[1] = X[1]; X1[2] = X[2]; }
                if (r == 2) { X2[0] = X[0]; X2[1] = X[1]; X2[2] = X[2]; }
            }
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
                    const bool row_in = gy >= 0 && gy < LI.h;
#pragma unroll
                    for (int d = 0; d < ND; d++) {
                        const int gx = ipx + tcol[k] + 2 * d;
                        const bool in0 = row_in && gx >= 0 && gx < LI.w, in1 = row_in && gx + 1 >= 0 && gx + 1 < LI.w;
                        const uint32_t m = (in0 ? 0xffffu : 0u) | (in1 ? 0xffff0000u : 0u);
                        dxp[r][d] = as_v2s(as_u32(dxp[r][d]) & m);
                        dyp[r][d] = as_v2s(as_u32(dyp[r][d]) & m);
                    }
                }
            }
            static_for<S>([&](auto jj) {
                constexpr int j = jj;
                // I sample: source columns (j+1, j+2) of rows 1 and 2
                const v2s s1 = byte_pair<j + 1, 3>(X1), s2 = byte_pair<j + 1, 3>(X2);
                const int iv = dot2(s2, W1, dot2(s1, W0, 1 << (W_BITS - 6))) >> (W_BITS - 5);
                // derivative samples: derivative columns (j, j+1) of derivative rows 0 and 1
                v2s gx0, gx1, gy0, gy1;
                if constexpr (j % 2 == 0) {
                    gx0 = dxp[0][j / 2]; gx1 = dxp[1][j / 2]; gy0 = dyp[0][j / 2]; gy1 = dyp[1][j / 2];
                } else {
                    gx0 = pair_shift(dxp[0][j / 2], dxp[0][j / 2 + 1]); gx1 = pair_shift(dxp[1][j / 2], dxp[1][j / 2 + 1]);
                    gy0 = pair_shift(dyp[0][j / 2], dyp[0][j / 2 + 1]); gy1 = pair_shift(dyp[1][j / 2], dyp[1][j / 2 + 1]);
                }
                int ixv = dot2(gx1, W1, dot2(gx0, W0, 1 << (W_BITS - 1))) >> W_BITS;
                int iyv = dot2(gy1, W1, dot2(gy0, W0, 1 << (W_BITS - 1))) >> W_BITS;
                int ineg = (1 << (W_BITS - 6)) - (iv << (W_BITS - 5));
                if constexpr (WW % S != 0) {
                    const bool on = j < tlen[k];
                    ixv = on ? ixv : 0;
                    iyv = on ? iyv : 0;
                } else if constexpr (C::NTASK % 64 != 0) {
                    const bool on = tlen[k] != 0;
                    ixv = on ? ixv : 0;
                    iyv = on ? iyv : 0;
                }
                Ineg[k][j] = ineg;
                Ixv[k][j] = ixv;
                Iyv[k][j] = iyv;
                a11 += __mul24(ixv, ixv);
                a12 += __mul24(ixv, iyv);
                a22 += __mul24(iyv, iyv);
            });
        }
        // |Ix*Ix| <= 4080^2 per pixel: 16-lane sums fit int32 while a lane holds <= 8 pixels
        const float A11 = i64_to_float(sum_pick<kSmall, 16>(a11)) * FLT_SCALE;
        const float A12 = i64_to_float(sum_pick<kSmall, 16>(a12)) * FLT_SCALE;
        const float A22 = i64_to_float(sum_pick<kSmall, 16>(a22)) * FLT_SCALE;
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
                if (tile_inside(LJ, jx0, jy0, C::JTW, C::JTH)) {
                    TileRegs<C::JPD, C::JTH> tj;
                    tile_issue(tj, LJ, jx0, jy0, lane);
                    tile_commit(tj, ldsJ, lane);
                } else {
                    tile_border<C::JPD, C::JTW, C::JTH>(ldsJ, LJ, jx0, jy0, lane);
                }
                __syncthreads();
                staged = true;
            }
            const Weights wj = bilinear_weights(nx - (float)inx, ny - (float)iny);
            const v2s V0 = {(short)wj.w00, (short)wj.w01}, V1 = {(short)wj.w10, (short)wj.w11};
            const int joff = (jx0 & 3) + (inx - jx0);
            const uint32_t* jrow0 = ldsJ + (iny - jy0) * C::JPD;
            int b1 = 0, b2 = 0;
#pragma unroll
            for (int k = 0; k < C::TPL; k++) {
                constexpr int NXJ = (S + 1 + 3) / 4;
                uint32_t Y0[NXJ], Y1[NXJ];
                row_dwords<NXJ>(jrow0 + trow[k] * C::JPD, joff + tcol[k], Y0);
                row_dwords<NXJ>(jrow0 + (trow[k] + 1) * C::JPD, joff + tcol[k], Y1);
                static_for<S>([&](auto qq) {
                    constexpr int q = qq;
                    // ((J bilinear + 256) >> 9) - I, with 256 - (I << 9) as the accumulator seed
                    const int diff = dot2(byte_pair<q, NXJ>(Y1), V1, dot2(byte_pair<q, NXJ>(Y0), V0, Ineg[k][q])) >> (W_BITS - 5);
                    b1 += __mul24(diff, Ixv[k][q]);
                    b2 += __mul24(diff, Iyv[k][q]);
                });
            }
            // |diff*Ix| <= 8160*4080 per pixel: 8-lane sums fit int32 while a lane holds <= 8 pixels
            const float fb1 = i64_to_float(sum_pick<kSmall, 8>(b1)) * FLT_SCALE;
            const float fb2 = i64_to_float(sum_pick<kSmall, 8>(b2)) * FLT_SCALE;
            const float dx = __fmul_rn(__fsub_rn(__fmul_rn(A12, fb2), __fmul_rn(A22, fb1)), D);
            const float dy = __fmul_rn(__fsub_rn(__fmul_rn(A12, fb1), __fmul_rn(A11, fb2)), D);
            nx = __fadd_rn(nx, dx); ny = __fadd_rn(ny, dy);
            sx = __fadd_rn(nx, half_x); sy = __fadd_rn(ny, half_y);
            if (__dadd_rn(__dmul_rn((double)dx, (double)dx), __dmul_rn((double)dy, (double)dy)) <= P.eps2) break;
            if (j > 0 && fabs((double)__fadd_rn(dx, pdx)) < 0.01 && fabs((double)__fadd_rn(dy, pdy)) < 0.01) {
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
            if (!staged || !tile_covers(jx0, jy0, iqx, iqy)) {
                jx0 = iqx - R; jy0 = iqy - R;
                __syncthreads();
                if (tile_inside(LJ, jx0, jy0, C::JTW, C::JTH)) {
                    TileRegs<C::JPD, C::JTH> tj;
                    tile_issue(tj, LJ, jx0, jy0, lane);
                    tile_commit(tj, ldsJ, lane);
                } else {
                    tile_border<C::JPD, C::JTW, C::JTH>(ldsJ, LJ, jx0, jy0, lane);
                }
                __syncthreads();
                staged = true;
            }
            const Weights we = bilinear_weights(qx - (float)iqx, qy - (float)iqy);
            const v2s V0 = {(short)we.w00, (short)we.w01}, V1 = {(short)we.w10, (short)we.w11};
            const int joff = (jx0 & 3) + (iqx - jx0);
            const uint32_t* jrow0 = ldsJ + (iqy - jy0) * C::JPD;
            int es = 0;
#pragma unroll
            for (int k = 0; k < C::TPL; k++) {
                constexpr int NXJ = (S + 1 + 3) / 4;
                uint32_t Y0[NXJ], Y1[NXJ];
                row_dwords<NXJ>(jrow0 + trow[k] * C::JPD, joff + tcol[k], Y0);
                row_dwords<NXJ>(jrow0 + (trow[k] + 1) * C::JPD, joff + tcol[k], Y1);
                static_for<S>([&](auto qq) {
                    constexpr int q = qq;
                    const int diff = dot2(byte_pair<q, NXJ>(Y1), V1, dot2(byte_pair<q, NXJ>(Y0), V0, Ineg[k][q])) >> (W_BITS - 5);
                    es += q < tlen[k] ? (diff < 0 ? -diff : diff) : 0;
                });
            }
            const float errval = i64_to_float(sum_pick<kSmall, 8>(es));
            Rz.err = __fdiv_rn(__fmul_rn(errval, 1.f), (float)(32 * WW * WH));
        }
    }
    Rz.x = sx;
    Rz.y = sy;
    return Rz;
}

template <int WW, int WH, bool FB>
__global__ __launch_bounds__(64, (Cfg<WW, WH>::TPL == 1 ? 4 : 2)) void k_lk_fast(Pyramid PI, Pyramid PJ, LKBuffers B, int n, LKParams P)
{
    using C = Cfg<WW, WH>;
    __shared__ uint32_t lds[C::LDS_DW];
    const int f = launch_slot(B, blockIdx.x, B.n_dev ? *B.n_dev : n);
    if (f < 0) return;
    if (B.seg_alive && !B.seg_alive[f]) return;
    const int lane = threadIdx.x;
    uint32_t* ldsI = lds;
    uint32_t* ldsJ = lds + C::I_DW;
    const float p0x = B.p_in[2 * f], p0y = B.p_in[2 * f + 1];
    const TrackResult r1 = track_point_fast<WW, WH>(PI, PJ, p0x, p0y, P, ldsI, ldsJ, lane);
    if (lane == 0) {
        if (B.p_fwd) { B.p_fwd[2 * f] = r1.x; B.p_fwd[2 * f + 1] = r1.y; }
        if (B.st_fwd) B.st_fwd[f] = (uint8_t)r1.status;
        if (B.err_fwd) B.err_fwd[f] = r1.err;
    }
    if (FB) {
        const TrackResult r2 = track_point_fast<WW, WH>(PJ, PI, r1.x, r1.y, P, ldsI, ldsJ, lane);
        if (lane == 0) {
            if (B.p_bwd) { B.p_bwd[2 * f] = r2.x; B.p_bwd[2 * f + 1] = r2.y; }
            if (B.st_bwd) B.st_bwd[f] = (uint8_t)r2.status;
            if (B.err_bwd) B.err_bwd[f] = r2.err;
            const float d = fb_distance(p0x, p0y, r2.x, r2.y, P.dist_form);
            if (B.dist) B.dist[f] = d;
            if (B.valid) B.valid[f] = d < P.fb_thr ? 1 : 0;
            if (B.seg_alive) seg_append(B, f, r1.x, r1.y, d, d < P.fb_thr);
        }
    }
}

template <int WW, int WH>
void launch_fast(hipStream_t s, const Pyramid& I, const Pyramid& J, const LKBuffers& B, int n, const LKParams& P,
                 bool fb)
{
    if (fb) hipLaunchKernelGGL((k_lk_fast<WW, WH, true>), dim3(B.order ? (n + 7) & ~7 : n), dim3(64), 0, s, I, J, B, n, P);
    else hipLaunchKernelGGL((k_lk_fast<WW, WH, false>), dim3(B.order ? (n + 7) & ~7 : n), dim3(64), 0, s, I, J, B, n, P);
}

}  // namespace

// Returns true when a specialised kernel exists for this window (and INITIAL_FLOW is not requested).
bool launch_lk_fast(hipStream_t s, const Pyramid& I, const Pyramid& J, const LKBuffers& B, int n, const LKParams& P,
                    bool fb)
{
    if (P.flags & ICELK_FLAG_INITIAL_FLOW) return false;
    if (P.win_w == 21 && P.win_h == 21) launch_fast<21, 21>(s, I, J, B, n, P, fb);
    else if (P.win_w == 31 && P.win_h == 31) launch_fast<31, 31>(s, I, J, B, n, P, fb);
    else if (P.win_w == 35 && P.win_h == 35) launch_fast<35, 35>(s, I, J, B, n, P, fb);
    else if (P.win_w == 15 && P.win_h == 15) launch_fast<15, 15>(s, I, J, B, n, P, fb);
    else return false;
    return true;
}

}  // namespace icelk
