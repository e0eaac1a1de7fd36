// k_utm.hip -- projection of finished tracks to map coordinates and the plausibility filter: the step right after
// the tracking loop.
//
// Replaces, per .npz of tracks, the triple Python loop of s2_cam_to_utm.py:243-347:
//   s2:243-255  every vertex: cropped -> uncropped photo coordinates (imports/camtools.py:414-421) and
//               Camera.photo_to_utm (imports/camtools.py:286-332): intersection of the pixel's view ray with the
//               sea-level plane, Krimmel & Rasmussen eq. 7 / 11
//   s2:282-291  u, v = vertex difference / tracking interval [m/s], speed = np.hypot(u, v)
//   s2:313-347  criteria 1-3: mean / max speed, speed ratio and direction change of consecutive vectors
// The hour bookkeeping around it (s2:257-278, 293-311, 349-363) is host logic (utm.py).
//
// One thread per track, everything in float64 with the reference's operation order (the library is built with
// -ffp-contract=off; '/' and sqrt are the correctly rounded device forms).  The nine direction cosines are inputs:
// the host forms them with numpy exactly as camtools.py:300-316 does, so no trigonometry is evaluated here except
// the acos of criterion 3.  A track is streamed vertex by vertex -- nothing is indexed dynamically, so the
// per-track state stays in registers.  Memory: 8 B per vertex in, 40 B per vector + 1 B per track out; the kernel
// is launch-latency bound at the reference's sizes (1e4 tracks x 3 vertices = 1 MB).
#include "icelk_internal.h"

namespace icelk {

namespace {

struct UtmPoint {
    double x, y;
};

__device__ __forceinline__ UtmPoint photo_to_utm(const UtmCamera& c, float fx, float fy)
{
    const double x = (double)fx + c.crop_left, y = (double)fy + c.crop_top;
    const double xi = x - c.half_w, yi = y - c.half_h;
    const double den = c.sigma * c.X[2] + xi * c.U[2] + yi * c.V[2];
    UtmPoint p;
    p.x = c.H * (c.sigma * c.X[0] + xi * c.U[0] + yi * c.V[0]) / den + c.E;
    p.y = c.H * (c.sigma * c.X[1] + xi * c.U[1] + yi * c.V[1]) / den + c.N;
    return p;
}

// np.hypot is the C library's hypot; this is the generic (non-FMA) algorithm of glibc 2.35
// (sysdeps/ieee754/dbl-64/e_hypot.c), which is what numpy calls on the reference's platform.  The exponent
// rescaling branches of the original (|x| > 2^511, |y| < 2^-459) cannot be reached by speeds in m/s; such inputs
// take the plain formula.
__device__ __forceinline__ double hypot_ref(double x, double y)
{
    double ax = fabs(x), ay = fabs(y);
    if (isinf(ax) || isinf(ay)) return HUGE_VAL;
    if (ax != ax || ay != ay) return ax + ay;
    if (ax < ay) { const double t = ax; ax = ay; ay = t; }
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

struct TrackState {
    UtmPoint prev;           // UTM position of the previous vertex
    double pu, pv, ps;       // previous vector
    double smax;             // max(speedsublist), Python semantics (NaN never replaces)
    double rmax, amax;       // max(speedratios), max(anglediffs)
    int k;                   // vectors so far
};

__global__ __launch_bounds__(256) void k_project_tracks(const float* __restrict__ tracks, int n, int nv, UtmCamera cam,
                                                        UtmFilter f, double* __restrict__ ox, double* __restrict__ oy,
                                                        double* __restrict__ ou, double* __restrict__ ov,
                                                        double* __restrict__ os, uint8_t* __restrict__ keep)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* t = tracks + (size_t)i * nv * 2;
    const int m = nv - 1;
    TrackState S;
    S.prev = photo_to_utm(cam, t[0], t[1]);
    S.pu = S.pv = S.ps = 0.0;
    S.smax = 0.0;
    S.rmax = S.amax = 0.0;
    S.k = 0;
    // next vector of the track: outputs, running maxima, the pair criteria against the previous vector
    auto vec = [&]() -> double {
        const int k = S.k;
        const UtmPoint q = photo_to_utm(cam, t[2 * (k + 1)], t[2 * (k + 1) + 1]);
        const double u = (q.x - S.prev.x) / f.interval_s, v = (q.y - S.prev.y) / f.interval_s;
        const double s = hypot_ref(u, v);
        const size_t o = (size_t)i * m + k;
        ox[o] = S.prev.x;
        oy[o] = S.prev.y;
        ou[o] = u;
        ov[o] = v;
        os[o] = s;
        if (k == 0) S.smax = s;
        else {
            if (s > S.smax) S.smax = s;
            const double dot = S.pu * u + S.pv * v;
            const double ang = fabs(acos(dot / (S.ps * s)) * (180.0 / 3.141592653589793238462643383279502884));
            const double hi = s > S.ps ? s : S.ps, lo = s < S.ps ? s : S.ps;   // max([s1, s2]), min([s1, s2])
            const double ratio = hi / lo;
            if (k == 1) { S.rmax = ratio; S.amax = ang; }
            else {
                if (ratio > S.rmax) S.rmax = ratio;
                if (ang > S.amax) S.amax = ang;
            }
        }
        S.prev = q;
        S.pu = u; S.pv = v; S.ps = s;
        S.k = k + 1;
        return s;
    };
    // np.mean(speedsublist) = (0 + pairwise sum) / m with numpy's pairwise sum: straight below 8 elements, else 8
    // accumulators over the full blocks of 8, a fixed tree over them, then the remainder one by one
    double sum;
    if (m < 8) {
        sum = 0.0;
        for (int k = 0; k < m; k++) sum += vec();
    } else {
        double r[8];
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = vec();
        for (int b = 1; b < m / 8; b++) {
#pragma unroll
            for (int j = 0; j < 8; j++) r[j] += vec();
        }
        sum = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (int k = 8 * (m / 8); k < m; k++) sum += vec();
    }
    if (m == 0) { keep[i] = 2; return; }       // max() of an empty list: the reference raises ValueError
    const double mean = (0.0 + sum) / (double)m;
    uint8_t kp = 1;
    if (mean < f.min_speed || S.smax > f.max_speed) kp = 0;
    else if (S.smax > f.speed_threshold) {
        if (m < 2) kp = 2;                     // max(speedratios) of an empty list
        else if (S.rmax > f.max_speedfactor) kp = 0;
        else if (S.amax > f.max_angle) kp = 0;
    }
    keep[i] = kp;
}

}  // namespace

void launch_project_tracks(hipStream_t s, const float* tracks, int n, int nv, const UtmCamera& cam, const UtmFilter& f,
                           double* x, double* y, double* u, double* v, double* speed, uint8_t* keep)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_project_tracks, dim3((n + 255) / 256), dim3(256), 0, s, tracks, n, nv, cam, f, x, y, u, v, speed,
                       keep);
}

}  // namespace icelk
