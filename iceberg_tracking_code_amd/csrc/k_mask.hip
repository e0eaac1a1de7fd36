// k_mask.hip -- the detector mask rasterised on the device.
//
// Replaces s1_lucaskanade_tracking.py:285-291: `cam.mask_meshgrid(x, y)` (imports/camtools.py:184-211) shifts the
// digitised water polygon by the crop offsets and asks matplotlib.path.Path.contains_points for every pixel
// centre of the frame; pixels inside become 255.  matplotlib's test (src/_path.h point_in_path_impl, radius 0) is
// a crossing-number test with a fixed tie rule, restated here operation for operation in double:
//   side(v) = (v.y >= ty);  an edge v0->v1 with side(v0) != side(v1) toggles the flag iff
//   ((v1.y - ty) * (v0.x - v1.x) >= (v1.x - tx) * (v0.y - v1.y)) == side(v1);
// the path closes from its last vertex to its first; fewer than 3 vertices contain nothing.
// One thread per pixel, the edge list is read with wave-uniform (scalar) loads; the 12 MB mask is written once
// and never crosses PCIe (the reference builds it on the host once per day, s1:285).
#include "icelk_internal.h"

namespace icelk {

namespace {

__global__ __launch_bounds__(256) void k_polygon_mask(const double* __restrict__ poly, int n, double crop_left,
                                                      double crop_top, int w, int h, uint8_t* __restrict__ mask,
                                                      int pitch)
{
    const int px = blockIdx.x * 64 + (threadIdx.x & 63), py = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (px >= w || py >= h) return;
    const double tx = (double)px, ty = (double)py;
    bool inside = false;
    if (n >= 3) {
        double x0 = poly[0] - crop_left, y0 = poly[1] - crop_top;
        const double sx = x0, sy = y0;
        bool f0 = y0 >= ty;
        for (int k = 1; k <= n; k++) {
            const double x1 = k < n ? poly[2 * k] - crop_left : sx;
            const double y1 = k < n ? poly[2 * k + 1] - crop_top : sy;
            const bool f1 = y1 >= ty;
            if (f0 != f1 && (((y1 - ty) * (x0 - x1) >= (x1 - tx) * (y0 - y1)) == f1)) inside = !inside;
            f0 = f1;
            x0 = x1;
            y0 = y1;
        }
    }
    mask[(size_t)py * pitch + px] = inside ? 255 : 0;
}

}  // namespace

void launch_polygon_mask(hipStream_t s, const double* poly, int n, double crop_left, double crop_top, int w, int h,
                         uint8_t* mask, int pitch)
{
    hipLaunchKernelGGL(k_polygon_mask, dim3((w + 63) / 64, (h + 3) / 4), dim3(256), 0, s, poly, n, crop_left, crop_top, w,
                       h, mask, pitch);
}

}  // namespace icelk
