/*
 * grid_oracle.c -- CPU restatement of the velocity gridding step (second half).
 *
 * TEST INFRASTRUCTURE ONLY (same rule as icelk_oracle.c).
 *
 * Follows s3_utm_to_gridded_utm.py:391-421: for every grid cell kept by trm.create_grid_across_fjord
 * (imports/tracking_misc.py:23-56: squares of `spacing` from the top-left corner of the fjord outline, kept when the
 * fjord polygon contains the cell centre) the velocities whose position lies in the cell are selected with
 * matplotlib.path.Path(poly).contains_points(points), and mean_u = np.sum(u_sel) / n, mean_v likewise,
 * speed = np.hypot(mean_u, mean_v).
 *
 * PARITY: tests/golden/grid_golden.npz holds (a) the grid the reference's own create_grid_across_fjord produced and
 * (b) per-cell results of the s3 loop body restated in the generator with the same third-party calls (matplotlib
 * contains_points, numpy sum / hypot) -- the s3 function itself needs a day folder tree, calibration workbook and
 * camera files and cannot be run here.  So (a) is pinned by the reference, (b) by the primitives it calls.
 *
 * Restated primitives: matplotlib's contains_points rule (see mask_oracle.c), numpy's pairwise summation
 * (numpy/core/src/umath/loops_utils.h.src: blocks of 128, 8 accumulators, halves aligned to 8) and glibc 2.35's
 * generic hypot (see utm_oracle.c).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

static int contains(const double* poly, int n, double tx, double ty)
{
    if (n < 3) return 0;
    int inside = 0;
    double x0 = poly[0], y0 = poly[1];
    int f0 = y0 >= ty;
    for (int k = 1; k <= n; k++) {
        const double x1 = k < n ? poly[2 * k] : poly[0];
        const double y1 = k < n ? poly[2 * k + 1] : poly[1];
        const int f1 = y1 >= ty;
        if (f0 != f1 && (((y1 - ty) * (x0 - x1) >= (x1 - tx) * (y0 - y1)) == f1)) inside ^= 1;
        f0 = f1;
        x0 = x1;
        y0 = y1;
    }
    return inside;
}

int orc_points_in_polygon(const double* poly, int n, const double* pts, int m, uint8_t* inside)
{
    if (n < 0 || m < 0) return -1;
    for (int i = 0; i < m; i++) inside[i] = (uint8_t)contains(poly, n, pts[2 * i], pts[2 * i + 1]);
    return 0;
}

static double pairwise(const double* a, size_t n)
{
    if (n < 8) {
        double r = 0.0;
        for (size_t i = 0; i < n; i++) r += a[i];
        return r;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; j++) r[j] = a[j];
        size_t i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    size_t n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise(a, n2) + pairwise(a + n2, n - n2);
}

static double hypot_glibc(double x, double y)
{
    double ax = fabs(x), ay = fabs(y);
    if (!(ax == ax) || !(ay == ay) || isinf(ax) || isinf(ay)) return hypot(x, y);
    if (ax < ay) { const double t = ax; ax = ay; ay = t; }
    if (ax > 0x1p+511 || (ay < 0x1p-459 && ay > ax * 0x1p-54)) return hypot(x, y);
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

/* Cells are indexed i * rows + j (i = column, j = row counted downwards from `top`), as the reference's double loop
 * walks them (tracking_misc.py:41-43); cell_on marks the cells create_grid_across_fjord kept.  For every kept cell:
 * count, and when count > 0 the means and the speed.  Cells that are off get count 0. */
int orc_grid_bin(const double* x, const double* y, const double* u, const double* v, int n, double left, double top,
                 double spacing, int cols, int rows, const uint8_t* cell_on, int* count, double* mean_u,
                 double* mean_v, double* speed)
{
    if (n < 0 || cols < 0 || rows < 0) return -1;
    double* bu = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    double* bv = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    if (!bu || !bv) { free(bu); free(bv); return -2; }
    for (int i = 0; i < cols; i++)
        for (int j = 0; j < rows; j++) {
            const int c = i * rows + j;
            count[c] = 0;
            mean_u[c] = mean_v[c] = speed[c] = 0.0;
            if (!cell_on[c]) continue;
            const double ox = left + i * spacing, oy = top - j * spacing;
            const double poly[8] = {ox, oy, ox + spacing, oy, ox + spacing, oy - spacing, ox, oy - spacing};
            int k = 0;
            for (int p = 0; p < n; p++)
                if (contains(poly, 4, x[p], y[p])) { bu[k] = u[p]; bv[k] = v[p]; k++; }
            count[c] = k;
            if (k > 0) {
                mean_u[c] = (0.0 + pairwise(bu, (size_t)k)) / (double)k;
                mean_v[c] = (0.0 + pairwise(bv, (size_t)k)) / (double)k;
                speed[c] = hypot_glibc(mean_u[c], mean_v[c]);
            }
        }
    free(bu);
    free(bv);
    return 0;
}
