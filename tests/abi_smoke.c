/* A plain-C caller of include/icelk.h: compiled (as C11) by tests/test_abi.py on the CPU, built and run against
 * libicelk.so by tests/test_gpu_api.py on the GPU.  It walks the loop body of s1_lucaskanade_tracking.py:307-450 once:
 * two frames in, corners of the first (s1:437), forward-backward track into the second (s1:323-333). */
#include <stdio.h>
#include <stdlib.h>

#include "icelk.h"

int main(void)
{
    const int w = 640, h = 480, max_corners = 200;
    icelk_t* ctx = NULL;
    if (icelk_create(0, w, h, 2, 4096, &ctx) != ICELK_OK) {
        fprintf(stderr, "icelk_create: %s\n", icelk_last_error(NULL));
        return 2;
    }
    int rc = icelk_synth_frame(ctx, 0, w, h, 0, 0, 1234);
    if (!rc) rc = icelk_synth_frame(ctx, 1, w, h, 300, -200, 1234);      /* texture origin moved by (300, -200) / 256 px */
    float* xy = (float*)malloc(sizeof(float) * 2 * max_corners);
    float* p1 = (float*)malloc(sizeof(float) * 2 * max_corners);
    float* dist = (float*)malloc(sizeof(float) * max_corners);
    uint8_t* valid = (uint8_t*)malloc(max_corners);
    int n = 0;
    if (!rc) rc = icelk_good_features(ctx, 0, 0, max_corners, 0.007, 10.0, 10, xy, max_corners, &n);
    if (!rc)
        rc = icelk_track_fb(ctx, 0, 1, xy, n, 21, 21, 3, ICELK_CRIT_COUNT | ICELK_CRIT_EPS, 30, 0.01, 1e-4, 1.0f, p1, NULL,
                            NULL, NULL, NULL, NULL, dist, valid);
    if (rc) {
        fprintf(stderr, "error %d: %s\n", rc, icelk_last_error(ctx));
        return 1;
    }
    int good = 0;
    double sx = 0, sy = 0;
    for (int i = 0; i < n; i++)
        if (valid[i]) {
            good++;
            sx += p1[2 * i] - xy[2 * i];
            sy += p1[2 * i + 1] - xy[2 * i + 1];
        }
    printf("corners %d valid %d mean_flow %.4f %.4f\n", n, good, good ? sx / good : 0.0, good ? sy / good : 0.0);
    free(xy); free(p1); free(dist); free(valid);
    icelk_destroy(ctx);
    return (n == max_corners && good > n / 2) ? 0 : 3;
}
