// k_grid.hip -- gridding of the projected velocities (second half).
//
// Replaces the per-cell loop of s3_utm_to_gridded_utm.py:391-421: for every square cell of the fjord grid
// (imports/tracking_misc.py:14-56) the reference runs matplotlib's Path(poly).contains_points over ALL velocities
// (cells x points tests), then mean_u = np.sum(u_sel) / n, mean_v likewise, speed = np.hypot(mean_u, mean_v).
//
// Here: one pass over the points finds the cells that contain each point -- only the 3 x 3 cells around its floor
// index can, and each is tested with matplotlib's own crossing rule on the cell's four vertices, formed as the
// reference forms them (left + i * spacing, top - j * spacing, + / - spacing), so points exactly on edges and corners
// land where contains_points puts them (in one, two or no cell).  The (cell, point index) pairs are sorted (rocPRIM
// radix sort, k_sort.hip), which restores the point order inside every cell, and one thread per cell adds its
// velocities in numpy's pairwise order (blocks of 128, 8 accumulators, halves aligned to 8) -- float64, bit for bit
// what np.sum gives -- and takes the hypot (glibc's algorithm, see k_utm.hip).
#include "icelk_internal.h"

namespace icelk {

namespace {

__device__ __forceinline__ bool contains_poly(const double* __restrict__ poly, int n, double tx, double ty)
{
    if (n < 3) return false;
    bool inside = false;
    double x0 = poly[0], y0 = poly[1];
    bool f0 = y0 >= ty;
    for (int k = 1; k <= n; k++) {
        const double x1 = k < n ? poly[2 * k] : poly[0];
        const double y1 = k < n ? poly[2 * k + 1] : poly[1];
        const bool f1 = y1 >= ty;
        if (f0 != f1 && (((y1 - ty) * (x0 - x1) >= (x1 - tx) * (y0 - y1)) == f1)) inside = !inside;
        f0 = f1;
        x0 = x1;
        y0 = y1;
    }
    return inside;
}

__global__ __launch_bounds__(256) void k_points_in_polygon(const double* __restrict__ poly, int n,
                                                           const double* __restrict__ pts, int m,
                                                           uint8_t* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    out[i] = contains_poly(poly, n, pts[2 * i], pts[2 * i + 1]) ? 1 : 0;
}

// square cell (i, j): [(x, y), (x + s, y), (x + s, y - s), (x, y - s)] with x = left + i * s, y = top - j * s
// (tracking_misc.py:14-21, 43)
__device__ __forceinline__ bool contains_cell(double left, double top, double s, int i, int j, double tx, double ty)
{
    const double x = left + i * s, y = top - j * s;
    const double p[8] = {x, y, x + s, y, x + s, y - s, x, y - s};
    return contains_poly(p, 4, tx, ty);
}

struct GridGeom {
    double left, top, spacing;
    int cols, rows;
};

// bit k = 3 * (dj + 1) + (di + 1) set: the point lies in cell (i0 + di, j0 + dj)
__device__ __forceinline__ unsigned hit_mask(const GridGeom& g, const uint8_t* __restrict__ cell_on, double tx,
                                             double ty, int* i0, int* j0)
{
    const double fi = floor((tx - g.left) / g.spacing), fj = floor((g.top - ty) / g.spacing);
    // far outside (or not finite): no cell
    if (!(fi >= -2.0 && fi <= (double)g.cols + 1.0 && fj >= -2.0 && fj <= (double)g.rows + 1.0)) {
        *i0 = *j0 = 0;
        return 0u;
    }
    *i0 = (int)fi;
    *j0 = (int)fj;
    unsigned m = 0;
    for (int dj = -1; dj <= 1; dj++)
        for (int di = -1; di <= 1; di++) {
            const int i = *i0 + di, j = *j0 + dj;
            if (i < 0 || j < 0 || i >= g.cols || j >= g.rows || !cell_on[i * g.rows + j]) continue;
            if (contains_cell(g.left, g.top, g.spacing, i, j, tx, ty)) m |= 1u << (3 * (dj + 1) + (di + 1));
        }
    return m;
}

__global__ __launch_bounds__(256) void k_grid_assign(const double* __restrict__ x, const double* __restrict__ y, int n,
                                                     GridGeom g, const uint8_t* __restrict__ cell_on,
                                                     unsigned long long* __restrict__ keys, int* __restrict__ key_count,
                                                     int key_cap)
{
    __shared__ int wave_tot[4];
    __shared__ int s_base;
    const int p = blockIdx.x * 256 + threadIdx.x;
    int i0 = 0, j0 = 0;
    const unsigned m = p < n ? hit_mask(g, cell_on, x[p], y[p], &i0, &j0) : 0u;
    const int cnt = __popc(m);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o);
        if (lane >= o) inc += v;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    int wbase = 0, total = 0;
    for (int k = 0; k < 4; k++) {
        wbase += k < wave ? wave_tot[k] : 0;
        total += wave_tot[k];
    }
    if (threadIdx.x == 0) s_base = total ? atomicAdd(key_count, total) : 0;
    __syncthreads();
    int at = s_base + wbase + inc - cnt;
    for (int k = 0; k < 9; k++)
        if (m & (1u << k)) {
            const int i = i0 + (k % 3) - 1, j = j0 + (k / 3) - 1;
            if (at < key_cap) keys[at] = ((unsigned long long)(unsigned)(i * g.rows + j) << 32) | (unsigned)p;
            at++;
        }
}

__device__ __forceinline__ double leaf_sum(const unsigned long long* __restrict__ keys, const double* __restrict__ a,
                                           int start, int n)
{
    auto at = [&](int t) { return a[(unsigned)keys[start + t]]; };
    if (n < 8) {
        double r = 0.0;
        for (int t = 0; t < n; t++) r += at(t);
        return r;
    }
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = at(j);
    int t = 8;
    for (; t < n - (n % 8); t += 8) {
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] += at(t + j);
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; t < n; t++) res += at(t);
    return res;
}

// numpy's pairwise sum over the velocities of keys[start, start + n), without recursion
__device__ double pairwise_sum(const unsigned long long* __restrict__ keys, const double* __restrict__ a, int start, int n)
{
    struct Frame { int start, n, stage; };
    Frame st[40];
    double vals[40];
    int fp = 0, sp = 0;
    st[fp++] = Frame{start, n, 0};
    while (fp) {
        Frame& f = st[fp - 1];
        if (f.n <= 128) {
            vals[sp++] = leaf_sum(keys, a, f.start, f.n);
            fp--;
            continue;
        }
        int n2 = f.n / 2;
        n2 -= n2 % 8;
        if (f.stage == 0) {
            f.stage = 1;
            st[fp++] = Frame{f.start, n2, 0};
        } else if (f.stage == 1) {
            f.stage = 2;
            st[fp++] = Frame{f.start + n2, f.n - n2, 0};
        } else {
            const double r = vals[sp - 2] + vals[sp - 1];
            sp -= 2;
            vals[sp++] = r;
            fp--;
        }
    }
    return vals[0];
}

__device__ __forceinline__ double hypot_np(double x, double y)   // see k_utm.hip hypot_ref
{
    double ax = fabs(x), ay = fabs(y);
    if (isinf(ax) || isinf(ay)) return HUGE_VAL;
    if (ax != ax || ay != ay) return ax + ay;
    if (ax < ay) { const double t = ax; ax = ay; ay = t; }
    if (ay <= ax * 0x1p-54) return ax + ay;
    double h = sqrt(ax * ax + ay * ay), t1, t2;
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

__device__ __forceinline__ int lower_bound(const unsigned long long* __restrict__ keys, int n, unsigned long long v)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(64) void k_grid_reduce(const unsigned long long* __restrict__ keys,
                                                    const int* __restrict__ key_count, const double* __restrict__ u,
                                                    const double* __restrict__ v, int ncells, int* __restrict__ count,
                                                    double* __restrict__ mean_u, double* __restrict__ mean_v,
                                                    double* __restrict__ speed)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= ncells) return;
    const int total = *key_count;
    const int b = lower_bound(keys, total, (unsigned long long)(unsigned)c << 32);
    const int e = lower_bound(keys, total, (unsigned long long)(unsigned)(c + 1) << 32);
    const int n = e - b;
    count[c] = n;
    double mu = 0.0, mv = 0.0, sp = 0.0;
    if (n > 0) {
        mu = (0.0 + pairwise_sum(keys, u, b, n)) / (double)n;
        mv = (0.0 + pairwise_sum(keys, v, b, n)) / (double)n;
        sp = hypot_np(mu, mv);
    }
    mean_u[c] = mu;
    mean_v[c] = mv;
    speed[c] = sp;
}

}  // namespace

void launch_points_in_polygon(hipStream_t s, const double* poly, int n, const double* pts, int m, uint8_t* out)
{
    if (m <= 0) return;
    hipLaunchKernelGGL(k_points_in_polygon, dim3((m + 255) / 256), dim3(256), 0, s, poly, n, pts, m, out);
}

void launch_grid_assign(hipStream_t s, const double* x, const double* y, int n, double left, double top, double spacing,
                        int cols, int rows, const uint8_t* cell_on, unsigned long long* keys, int* key_count, int key_cap)
{
    if (n <= 0) return;
    const GridGeom g{left, top, spacing, cols, rows};
    hipLaunchKernelGGL(k_grid_assign, dim3((n + 255) / 256), dim3(256), 0, s, x, y, n, g, cell_on, keys, key_count,
                       key_cap);
}

void launch_grid_reduce(hipStream_t s, const unsigned long long* keys, const int* key_count, const double* u,
                        const double* v, int ncells, int* count, double* mean_u, double* mean_v, double* speed)
{
    if (ncells <= 0) return;
    hipLaunchKernelGGL(k_grid_reduce, dim3((ncells + 63) / 64), dim3(64), 0, s, keys, key_count, u, v, ncells, count,
                       mean_u, mean_v, speed);
}

}  // namespace icelk
