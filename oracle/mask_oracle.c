/*
 * mask_oracle.c -- CPU restatement of the fjord-mask rasterisation.
 *
 * TEST INFRASTRUCTURE ONLY (same rule as icelk_oracle.c).
 *
 * PARITY PINNED: tests/golden/mask_golden.npz holds masks produced by the reference's own
 * Camera.mask_meshgrid (imports/camtools.py:184-211) called as s1_lucaskanade_tracking.py:285-291 does
 * (tests/golden/make_mask_golden.py); tests/test_oracle_mask.py checks this file against them byte for byte.
 *
 * The reference shifts the polygon by the crop offsets (camtools.py:189-194) and asks
 * matplotlib.path.Path(poly).contains_points(pixel centres) (camtools.py:207-208).  matplotlib is a third-party
 * dependency (not in /root/reference); its published algorithm (src/_path.h, point_in_path_impl, radius 0) is the
 * crossing-number test below: a +X ray from the point toggles a flag at every edge whose end points lie on
 * different sides of the point's y (side = "vertex y >= point y"), when
 *     ((y1 - ty) * (x0 - x1) >= (x1 - tx) * (y0 - y1)) == (y1 >= ty);
 * the polygon is closed implicitly from the last vertex to the first; fewer than 3 vertices contain nothing.
 * All arithmetic in double, in that order.  mask = 255 inside, 0 outside (s1:286-291).
 */
#include <stddef.h>
#include <stdint.h>

/* poly: n (x, y) pairs on the UNCROPPED photo; the mask covers pixel centres (0..w-1, 0..h-1) of the cropped
 * frame (origin upper left, as s1:290 asks). */
int orc_polygon_mask(const double* poly, int n, double crop_left, double crop_top, int w, int h, uint8_t* mask,
                     int stride)
{
    if (n < 0 || w <= 0 || h <= 0 || !mask || stride < w) return -1;
    for (int py = 0; py < h; py++) {
        const double ty = (double)py;
        for (int px = 0; px < w; px++) {
            const double tx = (double)px;
            int inside = 0;
            if (n >= 3) {
                double x0 = poly[0] - crop_left, y0 = poly[1] - crop_top;
                const double sx = x0, sy = y0;
                int f0 = y0 >= ty;
                for (int k = 1; k <= n; k++) {
                    const double x1 = k < n ? poly[2 * k] - crop_left : sx;
                    const double y1 = k < n ? poly[2 * k + 1] - crop_top : sy;
                    const int f1 = y1 >= ty;
                    if (f0 != f1 && (((y1 - ty) * (x0 - x1) >= (x1 - tx) * (y0 - y1)) == f1)) inside ^= 1;
                    f0 = f1;
                    x0 = x1;
                    y0 = y1;
                }
            }
            mask[(size_t)py * stride + px] = inside ? 255 : 0;
        }
    }
    return 0;
}
