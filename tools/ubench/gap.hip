// How much of a stream of back-to-back tracker-shaped launches (1-wave workgroups, 128 VGPRs, 8 KB LDS, 10^4
// workgroups of varying length) is lost to launch gaps and tails?  20 launches in ONE stream versus the same work as
// 2 x 20 half-size launches in TWO streams (each stream serial, the two independent of each other).
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(64, 4) void hog(int* out, int iters, int block0)
{
    __shared__ int lds[2048];
    lds[threadIdx.x] = threadIdx.x;
    asm volatile("v_mov_b32 v127, 0" ::: "v127");
    const int bid = blockIdx.x + block0;
    int a = threadIdx.x, b = bid;
    iters *= 1 + ((bid * 2654435761u) >> 29);
    for (int i = 0; i < iters; i++) { a = (a ^ b) + 0x1234567; b = (b ^ a) + 0x7654321; }
    out[(bid * 64 + threadIdx.x) & 0xffff] = a + b + lds[(a & 63)];
}

int main()
{
    int* d; hipMalloc(&d, 65536 * 4);
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEvent_t e0, e1, f0, f1;
    hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&f0); hipEventCreate(&f1);
    const int iters = 600, N = 20, WG = 10000;
    hog<<<WG, 64, 0, s1>>>(d, iters, 0);
    hipDeviceSynchronize();
    float one = 0, serial = 0, dual = 0;
    hipEventRecord(e0, s1); hog<<<WG, 64, 0, s1>>>(d, iters, 0); hipEventRecord(e1, s1); hipEventSynchronize(e1);
    hipEventElapsedTime(&one, e0, e1);
    hipEventRecord(e0, s1);
    for (int i = 0; i < N; i++) hog<<<WG, 64, 0, s1>>>(d, iters, 0);
    hipEventRecord(e1, s1); hipEventSynchronize(e1);
    hipEventElapsedTime(&serial, e0, e1);
    hipEventRecord(e0, s1); hipEventRecord(f0, s2);
    for (int i = 0; i < N; i++) {
        hog<<<WG / 2, 64, 0, s1>>>(d, iters, 0);
        hog<<<WG / 2, 64, 0, s2>>>(d, iters, WG / 2);
    }
    hipEventRecord(e1, s1); hipEventRecord(f1, s2);
    hipDeviceSynchronize();
    float a, b;
    hipEventElapsedTime(&a, e0, e1); hipEventElapsedTime(&b, f0, f1);
    dual = a > b ? a : b;
    printf("one launch %.1f us; %d launches in one stream %.1f us each; as two half-size streams %.1f us per pair\n",
           one * 1e3, N, serial * 1e3 / N, dual * 1e3 / N);
    return 0;
}
