/*
 * icelk_oracle.c -- CPU restatement of the sparse Lucas-Kanade tracking hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / the timed CPU baseline.  The shipped path is the HIP library in
 * iceberg_tracking_code_amd/csrc and it fails loudly when that library is missing.
 *
 * PARITY UNPINNED.  The reference (/root/reference) holds no arithmetic for this path: its
 * frame loop (s1_lucaskanade_tracking.py:307-450, s0_1_test_lucaskanade_tracking.py:77-181)
 * calls three third-party OpenCV functions,
 *     cv2.cvtColor(frame, cv2.COLOR_BGR2GRAY)          s1:283,311   s0_1:71,80
 *     cv2.calcOpticalFlowPyrLK(img0, img1, p0, None..) s1:323,326   s0_1:92,95
 *     cv2.goodFeaturesToTrack(frame_gray, mask=.., ..)  s1:437       s0_1:167
 * and OpenCV (pinned opencv=4.9.0 in environment.yml:254, 4.10.0 in s0_1.yml:199, "3.1.0"
 * in README.md:8) is neither vendored in the reference nor installed in this image, and the
 * reference ships no tests, fixtures or golden vectors for the path.  This file therefore
 * restates OpenCV's *published* algorithm (modules/imgproc color/pyramids/corner/featureselect
 * and modules/video lkpyramid, as summarised in SURVEY.md Appendix A) and is pinned only by
 * hand-computable known-answer vectors and analytic displacement tests (tests/test_oracle_*).
 *
 * Structure deliberately follows OpenCV's (level-major loop, materialised padded pyramids,
 * materialised Scharr derivative image) and NOT the GPU library's (feature-major, derivative
 * on the fly, reflect computed in the kernel), so that the two are independent statements of
 * the same arithmetic.
 *
 * Where OpenCV's own result is build dependent, one variant is fixed and named:
 *   - LK sums A11,A12,A22,b1,b2: exact integer accumulation (int64) converted to float once --
 *     the "acctype=int64 / itemtype=int" variant of lkpyramid.cpp; x86 builds accumulate in
 *     float lanes and differ from it by a few ULP.
 *   - err of points whose status is 0 is left at 0 (OpenCV leaves it uninitialised).
 *   - equal corner responses are ordered by higher raster address first (OpenCV >= 3.4).
 *   - the box filter sums each window row left-to-right and then the row sums top-to-bottom,
 *     both in double.  OpenCV keeps running double sums; for 8-bit input every term is a float
 *     product in [2^-27, 2^-3], so every such sum is EXACT in double and the order cannot matter
 *     (tests/test_oracle_kat.py::test_box_sums_of_the_covariance_planes_are_exact_in_double).
 *   - Sobel: separately rounded multiply and add in OpenCV's generic RowFilter / SymmColumnSmall
 *     operation order (a build without FMA contraction, e.g. 3.1 SSE2); an AVX2-dispatched 4.x
 *     build fuses them (v_muladd) and can differ in the last ulp of the eigenvalue map.
 *   - the forward-backward distance is the C library's hypotf, as np.hypot on float32 (s1:330);
 *     pinned against numpy (test_fb_distance_is_numpy_hypot_on_float32).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_PAR_MIN_PIXELS 500000L
#define ORC_OK 0
#define ORC_EARG -1
#define ORC_ENOMEM -2

/* cv flags / criteria bits (cv2.TERM_CRITERIA_COUNT=1, _EPS=2; s1:248) */
#define ORC_CRIT_COUNT 1
#define ORC_CRIT_EPS 2
#define ORC_FLAG_INITIAL_FLOW 4
#define ORC_FLAG_MIN_EIGENVALS 8

int orc_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------
 * Named variants of the build-dependent OpenCV semantics ("a named, unit-tested switch").
 * The DEFAULT of each is what the HIP kernels compute; the others exist to measure how far a differently built
 * OpenCV could be from it (tools/oracle_variants.py).  All are from knowledge of upstream
 * OpenCV's sources -- none can be checked against a cv2 binary here.
 *
 *   "lk_sums"   0  A11,A12,A22,b1,b2 accumulated exactly (int64), converted to float once  [default]
 *               1  OpenCV 3.x x86 SSE2 block of LKTrackerInvoker: the matrix sums in four float lanes over groups of 4
 *                  pixels of a window row (lane = x mod 4, every product exact in float, one rounding per add), the
 *                  pixels of a row beyond the last whole group in a scalar float accumulator, at the end
 *                  acc += ((l0 + l1) + l2) + l3; the residual sums b1,b2 in 2 x 4 lanes over groups of 8 pixels
 *                  (products converted int32 -> float first: they exceed 2^24), acc += (q0[k] + q1[k]) pairs
 *               2  OpenCV 4.x universal-intrinsic block (CV_SIMD128): as 1 but the matrix sums walk groups of 8
 *                  pixels (so a 21-px row leaves 5 pixels to the scalar accumulator instead of 1) and the lanes are
 *                  folded pairwise, (l0 + l2) + (l1 + l3) (v_reduce_sum)
 *   "sobel_fma" bit 0: the symmetric column pass of the scaled Sobel kernel fused, fmaf(r0 + r2, k1, r1 * k0)
 *                  (4.x SymmColumnSmallVec_32f's v_muladd in an FMA3 / AVX2-dispatched build)
 *               bit 1: the row pass k1*l + k0*c + k1*r with its two additions fused (a RowFilter loop contracted by
 *                  a compiler that targets FMA)
 *   "eig_fma"   1  calcMinEigenVal's (a-c)^2 + b^2 as fmaf(b, b, t*t) (4.x v_muladd(v_b, v_b, v_t * v_t))
 * ---------------------------------------------------------------------------------------- */
static int g_lk_sums = 0, g_sobel_fma = 0, g_eig_fma = 0;
int orc_set_variant(const char* name, int value)
{
    if (!name) return ORC_EARG;
    if (!strcmp(name, "lk_sums") && value >= 0 && value <= 2) { g_lk_sums = value; return ORC_OK; }
    if (!strcmp(name, "sobel_fma") && value >= 0 && value <= 3) { g_sobel_fma = value; return ORC_OK; }
    if (!strcmp(name, "eig_fma") && (value == 0 || value == 1)) { g_eig_fma = value; return ORC_OK; }
    return ORC_EARG;
}
int orc_get_variant(const char* name)
{
    if (!name) return ORC_EARG;
    if (!strcmp(name, "lk_sums")) return g_lk_sums;
    if (!strcmp(name, "sobel_fma")) return g_sobel_fma;
    if (!strcmp(name, "eig_fma")) return g_eig_fma;
    return ORC_EARG;
}

/* float-lane accumulators of the x86 SIMD blocks (lk_sums 1 / 2) */
typedef struct { float q[3][4]; float t[3]; } orc_lanes_a;   /* A11, A12, A22: four lanes + scalar accumulator */
typedef struct { float q0[4], q1[4]; float t[2]; } orc_lanes_b;

static void lanes_a_row(orc_lanes_a* L, const int16_t* dI, int win_w, int group)
{
    int x = 0;
    for (; x <= win_w - group; x += group)
        for (int k = 0; k < group; k++) {
            const float fx = (float)dI[2 * (x + k)], fy = (float)dI[2 * (x + k) + 1];
            float* a11 = &L->q[0][k & 3]; float* a12 = &L->q[1][k & 3]; float* a22 = &L->q[2][k & 3];
            *a22 = *a22 + fy * fy;
            *a12 = *a12 + fx * fy;
            *a11 = *a11 + fx * fx;
        }
    for (; x < win_w; x++) {
        const int ix = dI[2 * x], iy = dI[2 * x + 1];
        L->t[0] += (float)(ix * ix);
        L->t[1] += (float)(ix * iy);
        L->t[2] += (float)(iy * iy);
    }
}

static float lanes_a_fold(const orc_lanes_a* L, int which, int mode)
{
    const float* q = L->q[which];
    float acc = L->t[which];
    if (mode == 1) acc += ((q[0] + q[1]) + q[2]) + q[3];
    else acc += (q[0] + q[2]) + (q[1] + q[3]);
    return acc;
}

/* one pixel's residual x gradient products into the lane the SIMD block puts them in */
static void lanes_b_px(orc_lanes_b* L, int k, int diff, int ix, int iy)
{
    float* q = ((k >> 1) & 1) ? L->q1 : L->q0;      /* pixels 0,1,4,5 of a group of 8 -> q0; 2,3,6,7 -> q1 */
    const int lane = (k & 1) * 2;
    q[lane] = q[lane] + (float)(diff * ix);
    q[lane + 1] = q[lane + 1] + (float)(diff * iy);
}

/* BORDER_REFLECT_101:  ... 2 1 | 0 1 2 ... n-2 n-1 | n-2 n-3 ...    */
static int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        else p = 2 * n - 2 - p;
    }
    return p;
}

/* cvRound: nearest integer, ties to even (default FP rounding mode). */
static int round_half_even(float v) { return (int)lrintf(v); }
static int floor_int(float v) { return (int)floorf(v); }

/* ------------------------------------------------------------------------------------------
 * A.1 cvtColor(COLOR_BGR2GRAY), 8-bit.  Replaces cv2.cvtColor at s1:311.
 * variant 3: OpenCV 3.x  (c0*1868 + c1*9617 + c2*4899 + 2^13) >> 14
 * variant 4: OpenCV 4.x  (c0*3735 + c1*19235 + c2*9798 + 2^14) >> 15
 * c0 is array channel 0.  (The reference feeds PIL RGB arrays, so c0 is really R: s1:310-311.)
 * ---------------------------------------------------------------------------------------- */
int orc_bgr2gray(const uint8_t* src, int w, int h, int src_stride, uint8_t* dst, int dst_stride,
                 int variant)
{
    if (!src || !dst || w <= 0 || h <= 0) return ORC_EARG;
    int k0, k1, k2, sh;
    if (variant == 3) { k0 = 1868; k1 = 9617; k2 = 4899; sh = 14; }
    else if (variant == 4) { k0 = 3735; k1 = 19235; k2 = 9798; sh = 15; }
    else return ORC_EARG;
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * src_stride;
        uint8_t* d = dst + (size_t)y * dst_stride;
        for (int x = 0; x < w; x++)
            d[x] = (uint8_t)((s[3 * x] * k0 + s[3 * x + 1] * k1 + s[3 * x + 2] * k2 + (1 << (sh - 1))) >> sh);
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * A.3 pyrDown, 8-bit: separable [1 4 6 4 1], (sum + 128) >> 8, dst = ((w+1)/2, (h+1)/2).
 * ---------------------------------------------------------------------------------------- */
int orc_pyrdown(const uint8_t* src, int w, int h, int src_stride, uint8_t* dst, int dst_stride)
{
    if (!src || !dst || w <= 0 || h <= 0) return ORC_EARG;
    int dw = (w + 1) / 2, dh = (h + 1) / 2;
    int failed = 0;
    /* small images: a parallel region costs more than the work (128 spinning threads on the GPU box) */
#pragma omp parallel if ((long)w * h > ORC_PAR_MIN_PIXELS)
    {
    int* rows = (int*)malloc(sizeof(int) * (size_t)dw * 5);
    if (!rows) {
#pragma omp atomic write
        failed = 1;
    }
#pragma omp for schedule(static)
    for (int dy = 0; dy < dh; dy++) {
        if (!rows) continue;
        for (int k = 0; k < 5; k++) {
            int sy = reflect101(2 * dy - 2 + k, h);
            const uint8_t* s = src + (size_t)sy * src_stride;
            int* r = rows + (size_t)k * dw;
            for (int dx = 0; dx < dw; dx++) {
                int c = 2 * dx;
                r[dx] = s[reflect101(c - 2, w)] + s[reflect101(c + 2, w)] +
                        4 * (s[reflect101(c - 1, w)] + s[reflect101(c + 1, w)]) + 6 * s[reflect101(c, w)];
            }
        }
        uint8_t* d = dst + (size_t)dy * dst_stride;
        for (int dx = 0; dx < dw; dx++) {
            int v = rows[dx] + rows[4 * dw + dx] + 4 * (rows[dw + dx] + rows[3 * dw + dx]) + 6 * rows[2 * dw + dx];
            d[dx] = (uint8_t)((v + 128) >> 8);
        }
    }
    free(rows);
    }
    return failed ? ORC_ENOMEM : ORC_OK;
}

/* A.2 number of pyramid levels actually built: stop when the NEXT level would not exceed winSize. */
int orc_pyramid_levels(int w, int h, int win_w, int win_h, int max_level)
{
    int level;
    for (level = 0; level <= max_level; level++) {
        w = (w + 1) / 2;
        h = (h + 1) / 2;
        if (w <= win_w || h <= win_h) return level;
    }
    return max_level;
}

/* ------------------------------------------------------------------------------------------
 * A.4 Scharr derivative of an 8-bit image -> interleaved int16 (Ix, Iy).
 * ---------------------------------------------------------------------------------------- */
int orc_scharr(const uint8_t* src, int w, int h, int src_stride, int16_t* dst, int dst_stride_elems)
{
    if (!src || !dst || w <= 0 || h <= 0) return ORC_EARG;
    int failed = 0;
#pragma omp parallel if ((long)w * h > ORC_PAR_MIN_PIXELS)
    {
    int* t0 = (int*)malloc(sizeof(int) * (size_t)(w + 2) * 2);
    if (!t0) {
#pragma omp atomic write
        failed = 1;
    }
    int* t1 = t0 ? t0 + (w + 2) : NULL;
#pragma omp for schedule(static)
    for (int y = 0; y < h; y++) {
        if (!t0) continue;
        const uint8_t* r0 = src + (size_t)reflect101(y - 1, h) * src_stride;
        const uint8_t* r1 = src + (size_t)y * src_stride;
        const uint8_t* r2 = src + (size_t)reflect101(y + 1, h) * src_stride;
        for (int x = 0; x < w; x++) {
            t0[x + 1] = (r0[x] + r2[x]) * 3 + r1[x] * 10;
            t1[x + 1] = r2[x] - r0[x];
        }
        int xl = reflect101(-1, w), xr = reflect101(w, w);
        t0[0] = t0[xl + 1]; t1[0] = t1[xl + 1];
        t0[w + 1] = t0[xr + 1]; t1[w + 1] = t1[xr + 1];
        int16_t* d = dst + (size_t)y * dst_stride_elems;
        for (int x = 0; x < w; x++) {
            d[2 * x] = (int16_t)(t0[x + 2] - t0[x]);
            d[2 * x + 1] = (int16_t)((t1[x + 2] + t1[x]) * 3 + t1[x + 1] * 10);
        }
    }
    free(t0);
    }
    return failed ? ORC_ENOMEM : ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * Padded pyramid level (image + winSize border, reflect-101), as buildOpticalFlowPyramid keeps.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int w, h;       /* image size of this level */
    int pw;         /* padded row length (elements) */
    uint8_t* buf;   /* padded buffer */
    uint8_t* img;   /* pointer to pixel (0,0) inside buf */
} orc_level;

static int make_level(orc_level* L, int w, int h, int bw, int bh)
{
    L->w = w; L->h = h; L->pw = w + 2 * bw;
    L->buf = (uint8_t*)malloc((size_t)L->pw * (h + 2 * bh));
    if (!L->buf) return ORC_ENOMEM;
    L->img = L->buf + (size_t)bh * L->pw + bw;
    return ORC_OK;
}

static void pad_level(orc_level* L, int bw, int bh)
{
    /* rows inside the image first (left/right borders), then the rows above and below copy finished rows */
#pragma omp parallel for schedule(static) if ((long)L->w * L->h > ORC_PAR_MIN_PIXELS)
    for (int y = 0; y < L->h; y++) {
        uint8_t* d = L->img + (ptrdiff_t)y * L->pw;
        for (int x = 1; x <= bw; x++) {
            d[-x] = d[reflect101(-x, L->w)];
            d[L->w - 1 + x] = d[reflect101(L->w - 1 + x, L->w)];
        }
    }
    for (int y = -bh; y < L->h + bh; y++) {
        if (y >= 0 && y < L->h) continue;
        int sy = reflect101(y, L->h);
        uint8_t* d = L->img + (ptrdiff_t)y * L->pw;
        const uint8_t* s = L->img + (ptrdiff_t)sy * L->pw;
        if (y < 0 || y >= L->h)
            for (int x = 0; x < L->w; x++) d[x] = s[x];
        for (int x = 1; x <= bw; x++) {
            d[-x] = s[reflect101(-x, L->w)];
            d[L->w - 1 + x] = s[reflect101(L->w - 1 + x, L->w)];
        }
    }
}

static int build_pyramid(const uint8_t* img, int w, int h, int stride, int bw, int bh, int max_level,
                         orc_level* lv, int* nlev)
{
    int rc = make_level(&lv[0], w, h, bw, bh);
    if (rc) return rc;
    for (int y = 0; y < h; y++) memcpy(lv[0].img + (size_t)y * lv[0].pw, img + (size_t)y * stride, (size_t)w);
    pad_level(&lv[0], bw, bh);
    int built = 1;
    int eff = orc_pyramid_levels(w, h, bw, bh, max_level);
    for (int l = 1; l <= eff; l++) {
        int nw = (lv[l - 1].w + 1) / 2, nh = (lv[l - 1].h + 1) / 2;
        rc = make_level(&lv[l], nw, nh, bw, bh);
        if (rc) { *nlev = built; return rc; }
        built++;
        orc_pyrdown(lv[l - 1].img, lv[l - 1].w, lv[l - 1].h, lv[l - 1].pw, lv[l].img, lv[l].pw);
        pad_level(&lv[l], bw, bh);
    }
    *nlev = built;
    return ORC_OK;
}

static void free_pyramid(orc_level* lv, int n)
{
    for (int i = 0; i < n; i++) free(lv[i].buf);
}

/* Export the pyramid levels (unpadded) for the pyramid parity tests. out holds the levels
 * back to back, level l at row pitch w_l. */
int orc_build_pyramid(const uint8_t* img, int w, int h, int stride, int win_w, int win_h, int max_level,
                      uint8_t* out, int* out_levels)
{
    if (!img || !out || w <= 0 || h <= 0 || win_w <= 2 || win_h <= 2 || max_level < 0 || max_level > 30)
        return ORC_EARG;
    orc_level lv[32];
    int n = 0;
    int rc = build_pyramid(img, w, h, stride, win_w, win_h, max_level, lv, &n);
    if (rc == ORC_OK) {
        size_t off = 0;
        for (int l = 0; l < n; l++) {
            for (int y = 0; y < lv[l].h; y++)
                memcpy(out + off + (size_t)y * lv[l].w, lv[l].img + (size_t)y * lv[l].pw, (size_t)lv[l].w);
            off += (size_t)lv[l].w * lv[l].h;
        }
        if (out_levels) *out_levels = n - 1;
    }
    free_pyramid(lv, n);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * A.5/A.6 calcOpticalFlowPyrLK.  Replaces cv2.calcOpticalFlowPyrLK at s1:323,326.
 * next_pts is read only when flags has INITIAL_FLOW.  Returns the effective maxLevel (>= 0)
 * or a negative error.
 * ---------------------------------------------------------------------------------------- */
#define W_BITS 14

/* work counters of the last orc_pyrlk calls (ops accounting in DESIGN.md): [0] template patches
 * built (point x level visits that pass the bounds test), [1] LK iterations, [2] calls */
static long long g_lk_stats[3];
void orc_lk_stats(long long* out, int reset)
{
    for (int i = 0; i < 3; i++) { out[i] = g_lk_stats[i]; if (reset) g_lk_stats[i] = 0; }
}

static void bilinear_weights(float a, float b, int* w00, int* w01, int* w10, int* w11)
{
    *w00 = round_half_even((1.f - a) * (1.f - b) * (1 << W_BITS));
    *w01 = round_half_even(a * (1.f - b) * (1 << W_BITS));
    *w10 = round_half_even((1.f - a) * b * (1 << W_BITS));
    *w11 = (1 << W_BITS) - *w00 - *w01 - *w10;
}

static inline int descale(int v, int n) { return (v + (1 << (n - 1))) >> n; }

int orc_pyrlk(const uint8_t* prev, int prev_stride, const uint8_t* next, int next_stride, int w, int h,
              const float* prev_pts, float* next_pts, uint8_t* status, float* err, int n,
              int win_w, int win_h, int max_level, int crit_type, int max_count, double epsilon,
              int flags, double min_eig_threshold)
{
    if (!prev || !next || w <= 0 || h <= 0 || n < 0 || win_w <= 2 || win_h <= 2 || max_level < 0 || max_level > 30)
        return ORC_EARG;
    if (n == 0) return orc_pyramid_levels(w, h, win_w, win_h, max_level);
    if (!prev_pts || !next_pts || !status || !err) return ORC_EARG;

    orc_level pI[32], pJ[32];
    int nI = 0, nJ = 0;
    int rc = build_pyramid(prev, w, h, prev_stride, win_w, win_h, max_level, pI, &nI);
    if (rc == ORC_OK) rc = build_pyramid(next, w, h, next_stride, win_w, win_h, max_level, pJ, &nJ);
    if (rc != ORC_OK) { free_pyramid(pI, nI); free_pyramid(pJ, nJ); return rc; }
    int eff_level = nI - 1;

    if (!(crit_type & ORC_CRIT_COUNT)) max_count = 30;
    else max_count = max_count < 0 ? 0 : (max_count > 100 ? 100 : max_count);
    if (!(crit_type & ORC_CRIT_EPS)) epsilon = 0.01;
    else epsilon = epsilon < 0. ? 0. : (epsilon > 10. ? 10. : epsilon);
    epsilon *= epsilon;

    for (int i = 0; i < n; i++) { status[i] = 1; err[i] = 0.f; }

    const float min_eig_thr = (float)min_eig_threshold;
    const float half_x = (win_w - 1) * 0.5f, half_y = (win_h - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);

    /* derivative buffer for the largest level, padded by winSize with zeros */
    size_t dpw0 = (size_t)pI[0].w + 2 * win_w;
    int16_t* dbuf = (int16_t*)malloc(sizeof(int16_t) * 2 * dpw0 * ((size_t)pI[0].h + 2 * win_h));
    if (!dbuf) { free_pyramid(pI, nI); free_pyramid(pJ, nJ); return ORC_ENOMEM; }

    long long n_patches = 0, n_iters = 0;
    for (int level = eff_level; level >= 0; level--) {
        const orc_level* LI = &pI[level];
        const orc_level* LJ = &pJ[level];
        const int cols = LI->w, rows = LI->h;
        const int dpw = cols + 2 * win_w; /* padded derivative row, in (Ix,Iy) pairs */
        memset(dbuf, 0, sizeof(int16_t) * 2 * (size_t)dpw * (rows + 2 * win_h));
        int16_t* deriv = dbuf + 2 * ((size_t)win_h * dpw + win_w);
        orc_scharr(LI->img, cols, rows, LI->pw, deriv, 2 * dpw);
        const int dstep = 2 * dpw, stepI = LI->pw, stepJ = LJ->pw;

#pragma omp parallel if (n > 256)
        {
            int16_t* Iwin = (int16_t*)malloc(sizeof(int16_t) * 3 * (size_t)win_w * win_h);
            int16_t* dIwin = Iwin + (size_t)win_w * win_h;
#pragma omp for schedule(dynamic, 64) reduction(+ : n_patches, n_iters)
            for (int pt = 0; pt < n; pt++) {
                float px = prev_pts[2 * pt] * (float)(1. / (1 << level));
                float py = prev_pts[2 * pt + 1] * (float)(1. / (1 << level));
                float nx, ny;
                if (level == eff_level) {
                    if (flags & ORC_FLAG_INITIAL_FLOW) {
                        nx = next_pts[2 * pt] * (float)(1. / (1 << level));
                        ny = next_pts[2 * pt + 1] * (float)(1. / (1 << level));
                    } else { nx = px; ny = py; }
                } else {
                    nx = next_pts[2 * pt] * 2.f;
                    ny = next_pts[2 * pt + 1] * 2.f;
                }
                next_pts[2 * pt] = nx;
                next_pts[2 * pt + 1] = ny;

                px -= half_x; py -= half_y;
                int ipx = floor_int(px), ipy = floor_int(py);
                if (ipx < -win_w || ipx >= cols || ipy < -win_h || ipy >= rows) {
                    if (level == 0) { status[pt] = 0; err[pt] = 0.f; }
                    continue;
                }
                float a = px - ipx, b = py - ipy;
                int iw00, iw01, iw10, iw11;
                bilinear_weights(a, b, &iw00, &iw01, &iw10, &iw11);

                n_patches++;
                int64_t iA11 = 0, iA12 = 0, iA22 = 0;
                orc_lanes_a LA;
                memset(&LA, 0, sizeof LA);
                for (int y = 0; y < win_h; y++) {
                    const uint8_t* src = LI->img + (ptrdiff_t)(y + ipy) * stepI + ipx;
                    const int16_t* dsrc = deriv + (ptrdiff_t)(y + ipy) * dstep + (ptrdiff_t)ipx * 2;
                    int16_t* Ip = Iwin + (size_t)y * win_w;
                    int16_t* dIp = dIwin + (size_t)y * win_w * 2;
                    for (int x = 0; x < win_w; x++, dsrc += 2, dIp += 2) {
                        int ival = descale(src[x] * iw00 + src[x + 1] * iw01 + src[x + stepI] * iw10 +
                                           src[x + stepI + 1] * iw11, W_BITS - 5);
                        int ixval = descale(dsrc[0] * iw00 + dsrc[2] * iw01 + dsrc[dstep] * iw10 +
                                            dsrc[dstep + 2] * iw11, W_BITS);
                        int iyval = descale(dsrc[1] * iw00 + dsrc[3] * iw01 + dsrc[dstep + 1] * iw10 +
                                            dsrc[dstep + 3] * iw11, W_BITS);
                        Ip[x] = (int16_t)ival;
                        dIp[0] = (int16_t)ixval;
                        dIp[1] = (int16_t)iyval;
                        iA11 += (int64_t)ixval * ixval;
                        iA12 += (int64_t)ixval * iyval;
                        iA22 += (int64_t)iyval * iyval;
                    }
                    if (g_lk_sums) lanes_a_row(&LA, dIwin + (size_t)y * win_w * 2, win_w, g_lk_sums == 1 ? 4 : 8);
                }
                float A11 = (float)iA11 * FLT_SCALE;
                float A12 = (float)iA12 * FLT_SCALE;
                float A22 = (float)iA22 * FLT_SCALE;
                if (g_lk_sums) {
                    A11 = lanes_a_fold(&LA, 0, g_lk_sums) * FLT_SCALE;
                    A12 = lanes_a_fold(&LA, 1, g_lk_sums) * FLT_SCALE;
                    A22 = lanes_a_fold(&LA, 2, g_lk_sums) * FLT_SCALE;
                }
                float D = A11 * A22 - A12 * A12;
                float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) /
                               (float)(2 * win_w * win_h);
                if (flags & ORC_FLAG_MIN_EIGENVALS) err[pt] = minEig;
                if (minEig < min_eig_thr || D < FLT_EPSILON) {
                    if (level == 0) status[pt] = 0;
                    continue;
                }
                D = 1.f / D;
                nx -= half_x; ny -= half_y;
                float pdx = 0.f, pdy = 0.f;
                for (int j = 0; j < max_count; j++) {
                    int inx = floor_int(nx), iny = floor_int(ny);
                    if (inx < -win_w || inx >= cols || iny < -win_h || iny >= rows) {
                        if (level == 0) status[pt] = 0;
                        break;
                    }
                    n_iters++;
                    a = nx - inx; b = ny - iny;
                    bilinear_weights(a, b, &iw00, &iw01, &iw10, &iw11);
                    int64_t ib1 = 0, ib2 = 0;
                    orc_lanes_b LB;
                    memset(&LB, 0, sizeof LB);
                    const int whole8 = win_w & ~7;     /* pixels of a row the 8-wide SIMD block covers */
                    for (int y = 0; y < win_h; y++) {
                        const uint8_t* Jp = LJ->img + (ptrdiff_t)(y + iny) * stepJ + inx;
                        const int16_t* Ip = Iwin + (size_t)y * win_w;
                        const int16_t* dIp = dIwin + (size_t)y * win_w * 2;
                        for (int x = 0; x < win_w; x++, dIp += 2) {
                            int diff = descale(Jp[x] * iw00 + Jp[x + 1] * iw01 + Jp[x + stepJ] * iw10 +
                                               Jp[x + stepJ + 1] * iw11, W_BITS - 5) - Ip[x];
                            ib1 += (int64_t)diff * dIp[0];
                            ib2 += (int64_t)diff * dIp[1];
                            if (g_lk_sums) {
                                if (x < whole8) lanes_b_px(&LB, x & 7, diff, dIp[0], dIp[1]);
                                else { LB.t[0] += (float)(diff * dIp[0]); LB.t[1] += (float)(diff * dIp[1]); }
                            }
                        }
                    }
                    float b1 = (float)ib1 * FLT_SCALE;
                    float b2 = (float)ib2 * FLT_SCALE;
                    if (g_lk_sums) {
                        float sb[4];
                        for (int k = 0; k < 4; k++) sb[k] = LB.q0[k] + LB.q1[k];
                        b1 = (LB.t[0] + (sb[0] + sb[2])) * FLT_SCALE;
                        b2 = (LB.t[1] + (sb[1] + sb[3])) * FLT_SCALE;
                    }
                    float dx = (A12 * b2 - A22 * b1) * D;
                    float dy = (A12 * b1 - A11 * b2) * D;
                    nx += dx; ny += dy;
                    next_pts[2 * pt] = nx + half_x;
                    next_pts[2 * pt + 1] = ny + half_y;
                    if ((double)dx * dx + (double)dy * dy <= epsilon) break;
                    if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
                        next_pts[2 * pt] -= dx * 0.5f;
                        next_pts[2 * pt + 1] -= dy * 0.5f;
                        break;
                    }
                    pdx = dx; pdy = dy;
                }
                if (status[pt] && level == 0 && !(flags & ORC_FLAG_MIN_EIGENVALS)) {
                    float qx = next_pts[2 * pt] - half_x, qy = next_pts[2 * pt + 1] - half_y;
                    int iqx = floor_int(qx), iqy = floor_int(qy);
                    if (iqx < -win_w || iqx >= cols || iqy < -win_h || iqy >= rows) {
                        status[pt] = 0;
                        continue;
                    }
                    float aa = qx - iqx, bb = qy - iqy;
                    bilinear_weights(aa, bb, &iw00, &iw01, &iw10, &iw11);
                    int64_t esum = 0;
                    for (int y = 0; y < win_h; y++) {
                        const uint8_t* Jp = LJ->img + (ptrdiff_t)(y + iqy) * stepJ + iqx;
                        const int16_t* Ip = Iwin + (size_t)y * win_w;
                        for (int x = 0; x < win_w; x++) {
                            int diff = descale(Jp[x] * iw00 + Jp[x + 1] * iw01 + Jp[x + stepJ] * iw10 +
                                               Jp[x + stepJ + 1] * iw11, W_BITS - 5) - Ip[x];
                            esum += diff < 0 ? -diff : diff;
                        }
                    }
                    float errval = (float)esum;
                    err[pt] = errval * 1.f / (float)(32 * win_w * win_h);
                }
            }
            free(Iwin);
        }
    }
    g_lk_stats[0] += n_patches;
    g_lk_stats[1] += n_iters;
    g_lk_stats[2] += 1;
    free(dbuf);
    free_pyramid(pI, nI);
    free_pyramid(pJ, nJ);
    return eff_level;
}

/* ------------------------------------------------------------------------------------------
 * Forward-backward check of the reference loop (s1:323-333):
 *   p1 = LK(img0,img1,p0); p0r = LK(img1,img0,p1); dist = hypot(|p0-p0r|); valid = dist < 1
 * diff = abs(p0 - p0r) in float32, dist = np.hypot(diff_x, diff_y) on float32 (s1:329-330): numpy calls the
 * C library's hypotf, used here as it is (glibc evaluates it as (float)sqrt((double)x*x + (double)y*y));
 * tests/test_oracle_kat.py pins it against numpy, including distances one ulp either side of 1.0.
 * dist_form 1 = the demo script's float32 expression (dx**2 + dy**2)**0.5 (s0_1:99).
 * ---------------------------------------------------------------------------------------- */
static int g_dist_form = 0;
void orc_set_fb_distance(int form) { g_dist_form = form; }
float orc_fb_distance(float p0x, float p0y, float rx, float ry)
{
    float dx = fabsf(p0x - rx), dy = fabsf(p0y - ry);
    return g_dist_form == 1 ? sqrtf(dx * dx + dy * dy) : hypotf(dx, dy);
}

int orc_track_fb(const uint8_t* img0, int stride0, const uint8_t* img1, int stride1, int w, int h,
                 const float* p0, float* p1, float* p0r, uint8_t* st_fwd, uint8_t* st_bwd,
                 float* err_fwd, float* err_bwd, float* dist, uint8_t* valid, int n,
                 int win_w, int win_h, int max_level, int crit_type, int max_count, double epsilon,
                 double min_eig_threshold, float fb_threshold)
{
    int rc = orc_pyrlk(img0, stride0, img1, stride1, w, h, p0, p1, st_fwd, err_fwd, n, win_w, win_h,
                       max_level, crit_type, max_count, epsilon, 0, min_eig_threshold);
    if (rc < 0) return rc;
    rc = orc_pyrlk(img1, stride1, img0, stride0, w, h, p1, p0r, st_bwd, err_bwd, n, win_w, win_h,
                   max_level, crit_type, max_count, epsilon, 0, min_eig_threshold);
    if (rc < 0) return rc;
    for (int i = 0; i < n; i++) {
        dist[i] = orc_fb_distance(p0[2 * i], p0[2 * i + 1], p0r[2 * i], p0r[2 * i + 1]);
        valid[i] = dist[i] < fb_threshold ? 1 : 0;
    }
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * A.7 cornerMinEigenVal(image, blockSize, ksize=3) on an 8-bit image -> float map.
 * ---------------------------------------------------------------------------------------- */
int orc_min_eig_map(const uint8_t* img, int w, int h, int stride, int block_size, float* eig)
{
    if (!img || !eig || w <= 0 || h <= 0 || block_size <= 0) return ORC_EARG;
    double scale = (double)(1 << 2) * block_size;
    scale *= 255.0;
    scale = 1.0 / scale;
    const float k1 = (float)(1.0 * scale), k0 = (float)(2.0 * scale);

    size_t npx = (size_t)w * h;
    float* rdx = (float*)malloc(sizeof(float) * npx * 2); /* row-pass outputs */
    float* cov = (float*)malloc(sizeof(float) * npx * 3);
    double* rs = (double*)malloc(sizeof(double) * npx * 3); /* horizontal window sums */
    if (!rdx || !cov || !rs) { free(rdx); free(cov); free(rs); return ORC_ENOMEM; }
    float* rdy = rdx + npx;

    /* row pass: Dx uses [-1 0 1] (exact), Dy uses [k1 k0 k1] accumulated left to right */
#pragma omp parallel for schedule(static) if ((long)w * h > ORC_PAR_MIN_PIXELS)
    for (int y = 0; y < h; y++) {
        const uint8_t* s = img + (size_t)y * stride;
        for (int x = 0; x < w; x++) {
            int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
            rdx[(size_t)y * w + x] = (float)s[xp] - (float)s[xm];
            float t = k1 * (float)s[xm];
            if (g_sobel_fma & 2) {
                t = fmaf(k0, (float)s[x], t);
                t = fmaf(k1, (float)s[xp], t);
            } else {
                t = t + k0 * (float)s[x];
                t = t + k1 * (float)s[xp];
            }
            rdy[(size_t)y * w + x] = t;
        }
    }
    /* column pass + covariance products */
#pragma omp parallel for schedule(static) if ((long)w * h > ORC_PAR_MIN_PIXELS)
    for (int y = 0; y < h; y++) {
        int ym = reflect101(y - 1, h), yp = reflect101(y + 1, h);
        for (int x = 0; x < w; x++) {
            float dx;
            if (g_sobel_fma & 1)
                dx = fmaf(rdx[(size_t)ym * w + x] + rdx[(size_t)yp * w + x], k1, rdx[(size_t)y * w + x] * k0);
            else
                dx = (rdx[(size_t)ym * w + x] + rdx[(size_t)yp * w + x]) * k1 + rdx[(size_t)y * w + x] * k0;
            float dy = rdy[(size_t)yp * w + x] - rdy[(size_t)ym * w + x];
            float* c = cov + ((size_t)y * w + x) * 3;
            c[0] = dx * dx;
            c[1] = dx * dy;
            c[2] = dy * dy;
        }
    }
    /* unnormalised box filter, anchor = block_size/2, reflect-101, double sums */
    const int anchor = block_size / 2;
#pragma omp parallel for schedule(static) if ((long)w * h > ORC_PAR_MIN_PIXELS)
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            double s0 = 0, s1 = 0, s2 = 0;
            for (int k = 0; k < block_size; k++) {
                const float* c = cov + ((size_t)y * w + reflect101(x - anchor + k, w)) * 3;
                s0 += c[0]; s1 += c[1]; s2 += c[2];
            }
            double* r = rs + ((size_t)y * w + x) * 3;
            r[0] = s0; r[1] = s1; r[2] = s2;
        }
    }
#pragma omp parallel for schedule(static) if ((long)w * h > ORC_PAR_MIN_PIXELS)
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            double s0 = 0, s1 = 0, s2 = 0;
            for (int k = 0; k < block_size; k++) {
                const double* r = rs + ((size_t)reflect101(y - anchor + k, h) * w + x) * 3;
                s0 += r[0]; s1 += r[1]; s2 += r[2];
            }
            float a = (float)s0 * 0.5f, b = (float)s1, c = (float)s2 * 0.5f;
            const float t = a - c;
            eig[(size_t)y * w + x] = (a + c) - sqrtf(g_eig_fma ? fmaf(b, b, t * t) : t * t + b * b);
        }
    }
    free(rdx); free(cov); free(rs);
    return ORC_OK;
}

/* candidate = pointer into the eig map in OpenCV; here its raster index */
typedef struct { float v; int idx; } orc_cand;

static int cand_cmp(const void* pa, const void* pb)
{
    const orc_cand* a = (const orc_cand*)pa;
    const orc_cand* b = (const orc_cand*)pb;
    if (a->v > b->v) return -1;
    if (a->v < b->v) return 1;
    return a->idx > b->idx ? -1 : (a->idx < b->idx ? 1 : 0);
}

/* ------------------------------------------------------------------------------------------
 * A.7 goodFeaturesToTrack (Shi-Tomasi; useHarrisDetector=False).  Replaces s1:437.
 * out_xy holds up to cap (x,y) pairs; *out_n receives the number found (can exceed cap only
 * in the sense that the scan stops at cap).  max_corners <= 0 means "no limit".
 * ---------------------------------------------------------------------------------------- */
int orc_good_features(const uint8_t* img, int w, int h, int stride, const uint8_t* mask, int mask_stride,
                      int max_corners, double quality_level, double min_distance, int block_size,
                      float* out_xy, int cap, int* out_n)
{
    if (!img || !out_n || w <= 0 || h <= 0 || !(quality_level > 0) || min_distance < 0 || block_size <= 0)
        return ORC_EARG;
    *out_n = 0;
    size_t npx = (size_t)w * h;
    float* eig = (float*)malloc(sizeof(float) * npx);
    if (!eig) return ORC_ENOMEM;
    int rc = orc_min_eig_map(img, w, h, stride, block_size, eig);
    if (rc) { free(eig); return rc; }

    /* minMaxLoc over mask != 0 */
    int have = 0;
    float maxv = 0.f;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            if (mask && !mask[(size_t)y * mask_stride + x]) continue;
            float v = eig[(size_t)y * w + x];
            if (!have || v > maxv) { maxv = v; have = 1; }
        }
    double max_val = have ? (double)maxv : 0.0;
    float thresh = (float)(max_val * quality_level);

    /* threshold(TOZERO) + 3x3 dilate equality + mask, excluding the 1-px image border */
    orc_cand* cand = (orc_cand*)malloc(sizeof(orc_cand) * (npx ? npx : 1));
    if (!cand) { free(eig); return ORC_ENOMEM; }
    size_t total = 0;
    for (int y = 1; y < h - 1; y++)
        for (int x = 1; x < w - 1; x++) {
            float v = eig[(size_t)y * w + x];
            if (!(v > thresh)) continue; /* TOZERO keeps strictly greater; zeroed values never match */
            if (v == 0.f) continue;
            if (mask && !mask[(size_t)y * mask_stride + x]) continue;
            float m = 0.f; /* thresholded neighbours are >= 0 */
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    float q = eig[(size_t)(y + dy) * w + (x + dx)];
                    if (!(q > thresh)) q = 0.f;
                    if (q > m) m = q;
                }
            if (v == m) { cand[total].v = v; cand[total].idx = y * w + x; total++; }
        }
    free(eig);
    if (total == 0) { free(cand); return ORC_OK; }
    qsort(cand, total, sizeof(orc_cand), cand_cmp);

    int ncorners = 0;
    if (min_distance >= 1) {
        const int cell = (int)lrint(min_distance);
        const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
        /* per-cell singly linked lists of accepted corners */
        int* head = (int*)malloc(sizeof(int) * (size_t)gw * gh);
        int* nxt = (int*)malloc(sizeof(int) * total);
        float* ax = (float*)malloc(sizeof(float) * 2 * total);
        if (!head || !nxt || !ax) { free(head); free(nxt); free(ax); free(cand); return ORC_ENOMEM; }
        for (size_t i = 0; i < (size_t)gw * gh; i++) head[i] = -1;
        const double md2 = min_distance * min_distance;
        for (size_t i = 0; i < total; i++) {
            int y = cand[i].idx / w, x = cand[i].idx - y * w;
            int xc = x / cell, yc = y / cell;
            int x1 = xc - 1 < 0 ? 0 : xc - 1, y1 = yc - 1 < 0 ? 0 : yc - 1;
            int x2 = xc + 1 > gw - 1 ? gw - 1 : xc + 1, y2 = yc + 1 > gh - 1 ? gh - 1 : yc + 1;
            int good = 1;
            for (int yy = y1; yy <= y2 && good; yy++)
                for (int xx = x1; xx <= x2 && good; xx++)
                    for (int j = head[yy * gw + xx]; j >= 0; j = nxt[j]) {
                        float dx = x - ax[2 * j], dy = y - ax[2 * j + 1];
                        if ((double)(dx * dx + dy * dy) < md2) { good = 0; break; }
                    }
            if (good) {
                ax[2 * ncorners] = (float)x; ax[2 * ncorners + 1] = (float)y;
                nxt[ncorners] = head[yc * gw + xc];
                head[yc * gw + xc] = ncorners;
                if (out_xy && ncorners < cap) { out_xy[2 * ncorners] = (float)x; out_xy[2 * ncorners + 1] = (float)y; }
                ncorners++;
                if (max_corners > 0 && ncorners == max_corners) break;
                if (out_xy && ncorners == cap) break;
            }
        }
        free(head); free(nxt); free(ax);
    } else {
        for (size_t i = 0; i < total; i++) {
            int y = cand[i].idx / w, x = cand[i].idx - y * w;
            if (out_xy && ncorners < cap) { out_xy[2 * ncorners] = (float)x; out_xy[2 * ncorners + 1] = (float)y; }
            ncorners++;
            if (max_corners > 0 && ncorners == max_corners) break;
            if (out_xy && ncorners == cap) break;
        }
    }
    free(cand);
    *out_n = ncorners;
    return ORC_OK;
}
