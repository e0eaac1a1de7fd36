// k_lk.hip -- pyramidal Lucas-Kanade for gfx950: one 64-lane wavefront per feature, all pyramid
// levels (and, fused, the backward pass of the forward-backward check) in ONE launch.
//
// Replaces cv2.calcOpticalFlowPyrLK at s1_lucaskanade_tracking.py:323,326 (and the distance test
// of s1:329-333 in the fused form).  Arithmetic restated from OpenCV's LKTrackerInvoker (SURVEY.md
// A.5/A.6; OpenCV is not part of /root/reference), in the exact-integer-accumulator variant.
//
// MI355X mapping (DESIGN.md "K5"):
//   * OpenCV walks level-major over all points and materialises a 4 B/px Scharr image per level
//     (64 MB of HBM traffic per 12 MP frame).  Points are independent, so this kernel walks
//     feature-major: a wave carries its point from the coarsest level to level 0 and never
//     writes anything but the result.  The Scharr derivative is computed on the fly from the
//     (w+3)x(h+3) source patch staged in LDS.
//   * The template patch (I, Ix, Iy as int16) lives in REGISTERS: lane l owns window pixels
//     l, l+64, ...; the same pixels in every iteration, so the iteration loop reads only the
//     LDS-staged search region of the second image (a (w+1+2R)x(h+1+2R) u8 tile, R = margin)
//     and restages it only when the estimate leaves the tile.
//   * Ixx/Ixy/Iyy and the residual sums b1/b2 are exact integers: per-lane int32 partials
//     (|diff*Ix| <= 8160*4080, <= 64 px per lane), a 64-lane butterfly over (hi,lo) halves, one
//     int64 -> float conversion.  The result is independent of the lane layout and bit-identical
//     to the CPU oracle.
//   * A workgroup is one wave (64 threads): barriers are free, divergent iteration counts cost
//     only that wave's tail, and up to 32 features are resident per CU.
#include "lk_common.h"

namespace icelk {

bool launch_lk_fast(hipStream_t s, const Pyramid& I, const Pyramid& J, const LKBuffers& B, int n, const LKParams& P,
                    bool fb);  // k_lk_fast.hip
bool launch_lk_multi(hipStream_t s, const Pyramid& I, const Pyramid& J, const LKBuffers& B, int n, const LKParams& P,
                     bool fb);  // k_lk_multi.hip

namespace {

using namespace lk;

// LDS layout of one wave (offsets in bytes from the dynamic LDS base):
//   [0, tile_bytes)            : u8 tile -- first the (w+3)x(h+3) source patch of I, later the
//                                 (w+1+2R)x(h+1+2R) search region of J (same storage)
//   [deriv_off, +4*(w+1)(h+1)) : packed (Ix | Iy<<16) int16 pairs at the (w+1)x(h+1) integer positions
struct LdsPlan {
    int s_pitch;     // pitch of the I source patch
    int d_pitch;     // pitch (elements) of the derivative tile
    int j_pitch;     // pitch of the J search tile
    int j_w, j_h;    // J tile size
    int deriv_off;
};

__host__ __device__ inline LdsPlan make_plan(int win_w, int win_h, int margin)
{
    LdsPlan p;
    p.s_pitch = (win_w + 3 + 3) & ~3;
    p.d_pitch = win_w + 1;
    p.j_w = win_w + 1 + 2 * margin;
    p.j_h = win_h + 1 + 2 * margin;
    p.j_pitch = (p.j_w + 3) & ~3;
    int s_bytes = p.s_pitch * (win_h + 3);
    int j_bytes = p.j_pitch * p.j_h;
    int tile = s_bytes > j_bytes ? s_bytes : j_bytes;
    p.deriv_off = (tile + 15) & ~15;
    return p;
}

// Stage a w x h u8 tile whose top-left is image coordinate (x0, y0), reflect-101 outside.
__device__ __forceinline__ void stage_tile(uint8_t* lds, int pitch, int tw, int th, const Level& L, int x0, int y0,
                                           int lane)
{
    const int n = tw * th;
    const bool inside = x0 >= 0 && y0 >= 0 && x0 + tw <= L.w && y0 + th <= L.h;
    if (inside) {
        for (int i = lane; i < n; i += 64) {
            const int ty = i / tw, tx = i - ty * tw;
            lds[ty * pitch + tx] = L.ptr[(size_t)(y0 + ty) * L.pitch + (x0 + tx)];
        }
    } else {
        for (int i = lane; i < n; i += 64) {
            const int ty = i / tw, tx = i - ty * tw;
            const int sx = reflect101(x0 + tx, L.w), sy = reflect101(y0 + ty, L.h);
            lds[ty * pitch + tx] = L.ptr[(size_t)sy * L.pitch + sx];
        }
    }
}

// One direction of the tracker for one feature, executed by one wave.
//   PPL = window pixels per lane held in registers (ceil(win_w*win_h/64) <= PPL).
template <int PPL>
__device__ __forceinline__ TrackResult track_point(const Pyramid& PI, const Pyramid& PJ, float p0x, float p0y,
                                                   bool have_init, float initx, float inity,
                                                   const LKParams& P, uint8_t* lds, int lane)
{
    const int win_w = P.win_w, win_h = P.win_h;
    const int npx = win_w * win_h;
    const LdsPlan plan = make_plan(win_w, win_h, P.margin);
    uint8_t* tile = lds;
    uint32_t* dtile = reinterpret_cast<uint32_t*>(lds + plan.deriv_off);
    const float half_x = (win_w - 1) * 0.5f, half_y = (win_h - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);

    // window coordinates of this lane's pixels (same at every level)
    int wx[PPL], wy[PPL];
#pragma unroll
    for (int k = 0; k < PPL; k++) {
        const int idx = lane + 64 * k;
        const int yy = idx / win_w;
        wy[k] = yy;
        wx[k] = idx - yy * win_w;
    }

    TrackResult R;
    R.status = 1;
    R.err = 0.f;
    R.iters = 0;
    float sx = 0.f, sy = 0.f;  // the stored nextPts value
    // LKParams::sum_mode != 0: the sums in the float lanes of OpenCV's x86 SIMD blocks.  Every pixel's products go to LDS
    // in raster order; chain c < 4 of a plane adds the products of the columns x = c (mod 4) below `whole` row by row, chain
    // 4 those of the columns from `whole` on (the scalar tail of a row), each one float addition at a time in raster order
    // -- the order the SSE2 / CV_SIMD128 loops add them in (oracle/icelk_oracle.c lanes_a_row, lanes_b_px).  A lane per
    // chain; the five chains of a plane are folded as the blocks fold their lanes.
    float* fsum = reinterpret_cast<float*>(lds + plan.deriv_off + 4 * (win_w + 1) * (win_h + 1));
    auto lane_chain_sums = [&](int planes, int whole, bool pairwise, float (&out)[3]) {
        __syncthreads();
        float acc = 0.f;
        if (lane < 5 * planes) {
            const int p = lane / 5, c = lane - 5 * p;
            const float* src = fsum + p * npx;
            for (int y = 0; y < win_h; y++) {
                if (c < 4) for (int x = c; x < whole; x += 4) acc = __fadd_rn(acc, src[y * win_w + x]);
                else for (int x = whole; x < win_w; x++) acc = __fadd_rn(acc, src[y * win_w + x]);
            }
        }
#pragma unroll
        for (int p = 0; p < 3; p++) {
            if (p >= planes) break;
            const float q0 = __shfl(acc, 5 * p), q1 = __shfl(acc, 5 * p + 1), q2 = __shfl(acc, 5 * p + 2), q3 = __shfl(acc, 5 * p + 3);
            const float t = __shfl(acc, 5 * p + 4);
            out[p] = pairwise ? __fadd_rn(t, __fadd_rn(__fadd_rn(q0, q2), __fadd_rn(q1, q3)))
                              : __fadd_rn(t, __fadd_rn(__fadd_rn(__fadd_rn(q0, q1), q2), q3));
        }
        __syncthreads();
    };

    for (int level = P.top_level; level >= 0; level--) {
        const Level LI = PI.lv[level];
        const Level LJ = PJ.lv[level];
        const float scale = 1.f / (float)(1 << level);
        float px = p0x * scale, py = p0y * scale;
        if (level == P.top_level) {
            if (have_init) { sx = initx * scale; sy = inity * scale; }
            else { sx = px; sy = py; }
        } else {
            sx = sx * 2.f; sy = sy * 2.f;
        }
        px -= half_x; py -= half_y;
        const int ipx = (int)floorf(px), ipy = (int)floorf(py);
        if (ipx < -win_w || ipx >= LI.w || ipy < -win_h || ipy >= LI.h) {
            if (level == 0) { R.status = 0; R.err = 0.f; }
            continue;
        }
        const Weights wi = bilinear_weights(px - (float)ipx, py - (float)ipy);

        // ---- template patch: stage source, Scharr on the fly, bilinear into registers -----------
        __syncthreads();
        stage_tile(tile, plan.s_pitch, win_w + 3, win_h + 3, LI, ipx - 1, ipy - 1, lane);
        __syncthreads();
        {
            const int dn = (win_w + 1) * (win_h + 1);
            for (int i = lane; i < dn; i += 64) {
                const int ty = i / (win_w + 1), tx = i - ty * (win_w + 1);
                const int gx = ipx + tx, gy = ipy + ty;
                uint32_t packed = 0;
                if (gx >= 0 && gx < LI.w && gy >= 0 && gy < LI.h) {
                    const uint8_t* r0 = tile + ty * plan.s_pitch + tx;  // tile row ty = image row gy-1
                    const uint8_t* r1 = r0 + plan.s_pitch;
                    const uint8_t* r2 = r1 + plan.s_pitch;
                    const int t0l = (r0[0] + r2[0]) * 3 + r1[0] * 10;
                    const int t0r = (r0[2] + r2[2]) * 3 + r1[2] * 10;
                    const int t1l = r2[0] - r0[0], t1c = r2[1] - r0[1], t1r = r2[2] - r0[2];
                    const int ix = t0r - t0l;
                    const int iy = (t1r + t1l) * 3 + t1c * 10;
                    packed = ((uint32_t)ix & 0xffffu) | ((uint32_t)iy << 16);
                }
                dtile[ty * plan.d_pitch + tx] = packed;
            }
        }
        __syncthreads();

        int Ival[PPL];
        uint32_t dIval[PPL];  // (Ix | Iy << 16)
        int a11 = 0, a12 = 0, a22 = 0;
#pragma unroll
        for (int k = 0; k < PPL; k++) {
            Ival[k] = 0;
            dIval[k] = 0;
            if (lane + 64 * k < npx) {
                const uint8_t* s0 = tile + (wy[k] + 1) * plan.s_pitch + (wx[k] + 1);
                const uint8_t* s1 = s0 + plan.s_pitch;
                const int ival = descale(s0[0] * wi.w00 + s0[1] * wi.w01 + s1[0] * wi.w10 + s1[1] * wi.w11, W_BITS - 5);
                const uint32_t* d0 = dtile + wy[k] * plan.d_pitch + wx[k];
                const uint32_t* d1 = d0 + plan.d_pitch;
                const uint32_t q00 = d0[0], q01 = d0[1], q10 = d1[0], q11 = d1[1];
                const int ixv = descale((int)(short)(q00 & 0xffff) * wi.w00 + (int)(short)(q01 & 0xffff) * wi.w01 +
                                        (int)(short)(q10 & 0xffff) * wi.w10 + (int)(short)(q11 & 0xffff) * wi.w11, W_BITS);
                const int iyv = descale(((int)q00 >> 16) * wi.w00 + ((int)q01 >> 16) * wi.w01 +
                                        ((int)q10 >> 16) * wi.w10 + ((int)q11 >> 16) * wi.w11, W_BITS);
                Ival[k] = ival;
                dIval[k] = ((uint32_t)ixv & 0xffffu) | ((uint32_t)iyv << 16);
                a11 += ixv * ixv;
                a12 += ixv * iyv;
                a22 += iyv * iyv;
                if (P.sum_mode) {
                    const int idx = lane + 64 * k;
                    fsum[idx] = (float)(ixv * ixv);
                    fsum[npx + idx] = (float)(ixv * iyv);
                    fsum[2 * npx + idx] = (float)(iyv * iyv);
                }
            }
        }
        float A11 = (float)wave_sum_exact(a11) * FLT_SCALE;
        float A12 = (float)wave_sum_exact(a12) * FLT_SCALE;
        float A22 = (float)wave_sum_exact(a22) * FLT_SCALE;
        if (P.sum_mode) {
            // 3.x: groups of 4 pixels, lanes folded ((l0+l1)+l2)+l3; 4.x: groups of 8, folded (l0+l2)+(l1+l3)
            float o[3];
            lane_chain_sums(3, P.sum_mode == 1 ? (win_w & ~3) : (win_w & ~7), P.sum_mode == 2, o);
            A11 = __fmul_rn(o[0], FLT_SCALE); A12 = __fmul_rn(o[1], FLT_SCALE); A22 = __fmul_rn(o[2], FLT_SCALE);
        }
        float D = __fsub_rn(__fmul_rn(A11, A22), __fmul_rn(A12, A12));
        const float dif = __fsub_rn(A11, A22);
        const float rad = __fadd_rn(__fmul_rn(dif, dif), __fmul_rn(__fmul_rn(4.f, A12), A12));
        const float minEig = __fdiv_rn(__fsub_rn(__fadd_rn(A22, A11), sqrtf(rad)), (float)(2 * win_w * win_h));
        if (P.flags & ICELK_FLAG_MIN_EIGENVALS) R.err = minEig;
        if (minEig < P.min_eig_thr || D < 1.1920928955078125e-07f) {
            if (level == 0) R.status = 0;
            continue;
        }
        D = __fdiv_rn(1.f, D);

        // ---- iterations -------------------------------------------------------------------------
        float nx = sx - half_x, ny = sy - half_y;
        float pdx = 0.f, pdy = 0.f;
        int jx0 = 0, jy0 = 0;
        bool staged = false;
        for (int j = 0; j < P.max_count; j++) {
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (inx < -win_w || inx >= LJ.w || iny < -win_h || iny >= LJ.h) {
                if (level == 0) R.status = 0;
                break;
            }
            R.iters++;
            if (!staged || inx < jx0 || inx > jx0 + 2 * P.margin || iny < jy0 || iny > jy0 + 2 * P.margin) {
                jx0 = inx - P.margin; jy0 = iny - P.margin;
                __syncthreads();
                stage_tile(tile, plan.j_pitch, plan.j_w, plan.j_h, LJ, jx0, jy0, lane);
                __syncthreads();
                staged = true;
            }
            const Weights wj = bilinear_weights(nx - (float)inx, ny - (float)iny);
            const uint8_t* jbase = tile + (iny - jy0) * plan.j_pitch + (inx - jx0);
            int b1 = 0, b2 = 0;
#pragma unroll
            for (int k = 0; k < PPL; k++) {
                if (lane + 64 * k < npx) {
                    const uint8_t* j0 = jbase + wy[k] * plan.j_pitch + wx[k];
                    const uint8_t* j1 = j0 + plan.j_pitch;
                    const int diff = descale(j0[0] * wj.w00 + j0[1] * wj.w01 + j1[0] * wj.w10 + j1[1] * wj.w11,
                                             W_BITS - 5) - Ival[k];
                    b1 += diff * (int)(short)(dIval[k] & 0xffff);
                    b2 += diff * ((int)dIval[k] >> 16);
                    if (P.sum_mode) {
                        const int idx = lane + 64 * k;
                        fsum[idx] = (float)(diff * (int)(short)(dIval[k] & 0xffff));      // int32 -> float, as _mm_cvtepi32_ps
                        fsum[npx + idx] = (float)(diff * ((int)dIval[k] >> 16));
                    }
                }
            }
            float fb1 = (float)wave_sum_exact(b1) * FLT_SCALE;
            float fb2 = (float)wave_sum_exact(b2) * FLT_SCALE;
            if (P.sum_mode) {
                // both versions: groups of 8 pixels in 2 x 4 lanes; (q0[k] + q1[k]) pairs = chains (0 + 2) + (1 + 3)
                float o[3];
                lane_chain_sums(2, win_w & ~7, true, o);
                fb1 = __fmul_rn(o[0], FLT_SCALE); fb2 = __fmul_rn(o[1], FLT_SCALE);
            }
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

        // ---- residual error at level 0 -----------------------------------------------------------
        if (R.status && level == 0 && !(P.flags & ICELK_FLAG_MIN_EIGENVALS)) {
            const float qx = sx - half_x, qy = sy - half_y;
            const int iqx = (int)floorf(qx), iqy = (int)floorf(qy);
            if (iqx < -win_w || iqx >= LJ.w || iqy < -win_h || iqy >= LJ.h) {
                R.status = 0;
                continue;
            }
            if (!staged || iqx < jx0 || iqx > jx0 + 2 * P.margin || iqy < jy0 || iqy > jy0 + 2 * P.margin) {
                jx0 = iqx - P.margin; jy0 = iqy - P.margin;
                __syncthreads();
                stage_tile(tile, plan.j_pitch, plan.j_w, plan.j_h, LJ, jx0, jy0, lane);
                __syncthreads();
                staged = true;
            }
            const Weights we = bilinear_weights(qx - (float)iqx, qy - (float)iqy);
            const uint8_t* jbase = tile + (iqy - jy0) * plan.j_pitch + (iqx - jx0);
            int es = 0;
#pragma unroll
            for (int k = 0; k < PPL; k++) {
                if (lane + 64 * k < npx) {
                    const uint8_t* j0 = jbase + wy[k] * plan.j_pitch + wx[k];
                    const uint8_t* j1 = j0 + plan.j_pitch;
                    const int diff = descale(j0[0] * we.w00 + j0[1] * we.w01 + j1[0] * we.w10 + j1[1] * we.w11,
                                             W_BITS - 5) - Ival[k];
                    es += diff < 0 ? -diff : diff;
                }
            }
            const float errval = (float)wave_sum_exact(es);
            R.err = __fdiv_rn(__fmul_rn(errval, 1.f), (float)(32 * win_w * win_h));
        }
    }
    R.x = sx;
    R.y = sy;
    return R;
}

template <int PPL, bool FB>
__global__ __launch_bounds__(64) void k_lk(Pyramid PI, Pyramid PJ, LKBuffers B, int n, LKParams P)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int f = launch_slot(B, blockIdx.x, B.n_dev ? *B.n_dev : n);
    if (f < 0) return;
    if (B.seg_alive && !B.seg_alive[f]) return;
    const int lane = threadIdx.x;
    const float p0x = B.p_in[2 * f], p0y = B.p_in[2 * f + 1];
    const bool init = (P.flags & ICELK_FLAG_INITIAL_FLOW) != 0;
    float ix = 0.f, iy = 0.f;
    if (init) { ix = B.p_fwd[2 * f]; iy = B.p_fwd[2 * f + 1]; }
    const TrackResult r1 = track_point<PPL>(PI, PJ, p0x, p0y, init, ix, iy, P, lds, lane);
    if (lane == 0) {
        if (B.p_fwd) { B.p_fwd[2 * f] = r1.x; B.p_fwd[2 * f + 1] = r1.y; }
        if (B.st_fwd) B.st_fwd[f] = (uint8_t)r1.status;
        if (B.err_fwd) B.err_fwd[f] = r1.err;
        if (B.iters && !FB) B.iters[f] = (uint32_t)r1.iters;
    }
    if (FB) {
        const TrackResult r2 = track_point<PPL>(PJ, PI, r1.x, r1.y, false, 0.f, 0.f, P, lds, lane);
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
}

template <int PPL>
void launch_ppl(hipStream_t s, const Pyramid& I, const Pyramid& J, const LKBuffers& B, int n, const LKParams& P,
                bool fb, size_t lds)
{
    if (fb) hipLaunchKernelGGL((k_lk<PPL, true>), dim3(B.order ? (n + 15) & ~7 : n), dim3(64), lds, s, I, J, B, n, P);
    else hipLaunchKernelGGL((k_lk<PPL, false>), dim3(B.order ? (n + 15) & ~7 : n), dim3(64), lds, s, I, J, B, n, P);
}

}  // namespace

size_t lk_lds_bytes(const LKParams& P)
{
    const LdsPlan p = make_plan(P.win_w, P.win_h, P.margin);
    const size_t lane_sums = P.sum_mode ? 12u * (size_t)P.win_w * P.win_h : 0u;     // three planes of float products
    return (size_t)p.deriv_off + 4u * (size_t)(P.win_w + 1) * (P.win_h + 1) + lane_sums;
}

// Returns 0 or ICELK_EARG when the window needs more than 64 pixels per lane (> 4096 px).
int launch_lk(hipStream_t s, const Pyramid& I, const Pyramid& J, const LKBuffers& B, int n, const LKParams& P,
              bool fb)
{
    if (n <= 0) return ICELK_OK;
    // specialised windows: one feature per wave (k_lk_fast.hip) by default; ICELK_FLAG_MULTI_PER_WAVE selects the
    // several-features-per-wave form (k_lk_multi.hip: a third fewer vector instructions per feature, but 2-3 waves per
    // SIMD instead of 4 -- measured slower at 21x21, profiles/r02_lk_kernels.txt; kept as a third statement of the
    // arithmetic and for window sizes / chips where the balance tips)
    // (the "lk_sums" variants run in the window-specialised kernel too since round 4 -- k_lk_fast.hip chain_sums --, not in
    // the several-features-per-wave form)
    const bool generic = (P.flags & ICELK_FLAG_GENERIC_KERNEL) != 0;
    if ((P.flags & ICELK_FLAG_MULTI_PER_WAVE) && !generic && !P.sum_mode && launch_lk_multi(s, I, J, B, n, P, fb))
        return ICELK_OK;
    if (!generic && launch_lk_fast(s, I, J, B, n, P, fb)) return ICELK_OK;
    const int npx = P.win_w * P.win_h;
    const int ppl = (npx + 63) / 64;
    const size_t lds = lk_lds_bytes(P);
    if (lds > 64 * 1024) return ICELK_EARG;
    if (ppl <= 2) launch_ppl<2>(s, I, J, B, n, P, fb, lds);
    else if (ppl <= 4) launch_ppl<4>(s, I, J, B, n, P, fb, lds);
    else if (ppl <= 7) launch_ppl<7>(s, I, J, B, n, P, fb, lds);
    else if (ppl <= 10) launch_ppl<10>(s, I, J, B, n, P, fb, lds);
    else if (ppl <= 16) launch_ppl<16>(s, I, J, B, n, P, fb, lds);
    else if (ppl <= 20) launch_ppl<20>(s, I, J, B, n, P, fb, lds);
    else if (ppl <= 28) launch_ppl<28>(s, I, J, B, n, P, fb, lds);
    else if (ppl <= 40) launch_ppl<40>(s, I, J, B, n, P, fb, lds);
    else if (ppl <= 64) launch_ppl<64>(s, I, J, B, n, P, fb, lds);
    else return ICELK_EARG;
    return ICELK_OK;
}

}  // namespace icelk
