/*
 * utm_oracle.c -- CPU restatement of the track projection / plausibility filter that follows the tracking
 * loop.
 *
 * TEST INFRASTRUCTURE ONLY (same rule as icelk_oracle.c): only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.
 *
 * PARITY PINNED: unlike the tracking path this arithmetic lives in the reference itself (pure numpy), and
 * tests/golden/utm_golden.npz holds inputs and outputs produced by running the reference's own
 * s2_cam_to_utm.cam_to_utm() in the development container (tests/golden/make_utm_golden.py);
 * tests/test_oracle_utm.py checks this file against them bit for bit.
 *
 * Follows, per track (a row of the `tracks` array of one .npz, s2_cam_to_utm.py:233-234):
 *   s2_cam_to_utm.py:243-255   every vertex: cropped -> uncropped photo coordinates
 *                              (imports/camtools.py:414-421), then Camera.photo_to_utm
 *                              (imports/camtools.py:286-332, Krimmel & Rasmussen eq. 7 / 11)
 *   s2_cam_to_utm.py:282-291   u, v = vertex difference / tracking interval, speed = np.hypot(u, v)
 *   s2_cam_to_utm.py:313-315   criterion 1: np.mean(speed) < min_speed or max(speed) > max_speed
 *   s2_cam_to_utm.py:318-347   if max(speed) > speed_threshold: speed ratio and direction change of consecutive
 *                              vectors, criteria 2 and 3
 * The hour bookkeeping (s2:257-278,293-311,349-363) is host logic and is not here.
 *
 * Arithmetic notes (all float64, no contraction):
 *   - x, y come from float32 arrays via .tolist(): exact widening.
 *   - the nine direction cosines X, U, V are inputs: the caller forms them with numpy exactly as
 *     camtools.py:300-316 does, so no trigonometry is restated here.
 *   - np.hypot is the C library's hypot; glibc 2.35's generic (non-FMA) algorithm is restated below
 *     (checked equal to np.hypot on 2e5 random pairs in the development container).
 *   - np.mean of a short list = (0 + pairwise sum) / n with numpy's 8-accumulator pairwise sum.
 *   - Python's max()/min() keep the first element unless a later one compares greater/smaller (NaN never does).
 *   - np.arccos / np.degrees: libm acos, x * (180 / pi).  Only the comparison with max_angle uses them.
 * keep[i]: 1 kept, 0 dropped, 2 = the reference raises ValueError here (max() of an empty list: a track with
 * fewer than two vectors that exceeds speed_threshold, or no vector at all).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

typedef struct {
    double X[3], U[3], V[3];
    double sigma, H, E, N;
    double half_w, half_h;        /* pic['width'] / 2.0, pic['height'] / 2.0 */
    double crop_left, crop_top;
} orc_camera_t;

typedef struct {
    double interval_s, max_speed, min_speed, max_speedfactor, max_angle, speed_threshold;
} orc_utm_filter_t;

/* glibc 2.35 sysdeps/ieee754/dbl-64/e_hypot.c, generic kernel (no FMA); scaling branches for |x| > 2^511 and
 * |y| < 2^-459 are not needed for metres and are left to libm */
static double hypot_glibc(double x, double y)
{
    double ax = fabs(x), ay = fabs(y);
    if (!(ax == ax) || !(ay == ay) || isinf(ax) || isinf(ay)) return hypot(x, y);
    if (ax < ay) { const double t = ax; ax = ay; ay = t; }
    if (ax > 0x1p+511 || (ay < 0x1p-459 && ay > ax * 0x1p-54)) return hypot(x, y);
    if (ay <= ax * 0x1p-54) return ax + ay;
    double h = sqrt(ax * ax + ay * ay);
    double t1, t2;
    if (h <= 2.0 * ay) {
        const double delta = h - ay;
        t1 = ax * (2.0 * delta - ax);
        t2 = (delta - 2.0 * (ax - ay)) * delta;
    } else {
        const double delta = h - ax;
        t1 = 2.0 * delta * (ax - 2.0 * ay);
        t2 = (4.0 * delta - ay) * ay + delta * delta;
    }
    h -= (t1 + t2) / (2.0 * h);
    return h;
}

static double numpy_pairwise_sum(const double* a, int n)
{
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; i++) r += a[i];
        return r;
    }
    double r[8];
    for (int j = 0; j < 8; j++) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; j++) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

static void photo_to_utm(const orc_camera_t* c, double x, double y, double* tx, double* ty)
{
    x = x + c->crop_left;
    y = y + c->crop_top;
    const double xi = x - c->half_w, yi = y - c->half_h;
    const double den = c->sigma * c->X[2] + xi * c->U[2] + yi * c->V[2];
    *tx = c->H * (c->sigma * c->X[0] + xi * c->U[0] + yi * c->V[0]) / den + c->E;
    *ty = c->H * (c->sigma * c->X[1] + xi * c->U[1] + yi * c->V[1]) / den + c->N;
}

#define ORC_MAX_VEC 64

/* tracks: (n, nv, 2) float32.  x, y, u, v, speed: (n, nv-1) float64 (x, y = UTM position of the vector's first
 * vertex, as s2:296-297 stores it).  Returns 0, or -1 on bad arguments. */
int orc_project_tracks(const float* tracks, int n, int nv, const orc_camera_t* cam, const orc_utm_filter_t* f,
                       double* x, double* y, double* u, double* v, double* speed, uint8_t* keep)
{
    if (n < 0 || nv < 1 || nv - 1 > ORC_MAX_VEC || !cam || !f) return -1;
    const int m = nv - 1;
    for (int i = 0; i < n; i++) {
        const float* t = tracks + (size_t)i * nv * 2;
        double px, py, us[ORC_MAX_VEC], vs[ORC_MAX_VEC], ss[ORC_MAX_VEC];
        photo_to_utm(cam, (double)t[0], (double)t[1], &px, &py);
        for (int k = 1; k < nv; k++) {
            double qx, qy;
            photo_to_utm(cam, (double)t[2 * k], (double)t[2 * k + 1], &qx, &qy);
            us[k - 1] = (qx - px) / f->interval_s;
            vs[k - 1] = (qy - py) / f->interval_s;
            ss[k - 1] = hypot_glibc(us[k - 1], vs[k - 1]);
            x[(size_t)i * m + k - 1] = px;
            y[(size_t)i * m + k - 1] = py;
            u[(size_t)i * m + k - 1] = us[k - 1];
            v[(size_t)i * m + k - 1] = vs[k - 1];
            speed[(size_t)i * m + k - 1] = ss[k - 1];
            px = qx;
            py = qy;
        }
        if (m == 0) { keep[i] = 2; continue; }
        const double mean = (0.0 + numpy_pairwise_sum(ss, m)) / (double)m;
        double smax = ss[0];
        for (int k = 1; k < m; k++)
            if (ss[k] > smax) smax = ss[k];
        if (mean < f->min_speed || smax > f->max_speed) { keep[i] = 0; continue; }
        uint8_t kp = 1;
        if (smax > f->speed_threshold) {
            if (m < 2) { keep[i] = 2; continue; }
            double rmax = 0, amax = 0;
            for (int k = 0; k + 1 < m; k++) {
                const double dot = us[k] * us[k + 1] + vs[k] * vs[k + 1];
                const double m1 = hypot_glibc(us[k], vs[k]), m2 = hypot_glibc(us[k + 1], vs[k + 1]);
                const double ang = fabs(acos(dot / (m1 * m2)) * (180.0 / 3.141592653589793238462643383279502884));
                const double hi = ss[k + 1] > ss[k] ? ss[k + 1] : ss[k];   /* max([s1, s2]) */
                const double lo = ss[k + 1] < ss[k] ? ss[k + 1] : ss[k];   /* min([s1, s2]) */
                const double ratio = hi / lo;
                if (k == 0) { rmax = ratio; amax = ang; }
                else {
                    if (ratio > rmax) rmax = ratio;
                    if (ang > amax) amax = ang;
                }
            }
            if (rmax > f->max_speedfactor) kp = 0;
            else if (amax > f->max_angle) kp = 0;
        }
        keep[i] = kp;
    }
    return 0;
}
