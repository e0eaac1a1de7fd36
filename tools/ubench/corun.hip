// Co-scheduling microbenchmark: how long does a small kernel on a high-priority stream take while a register-
// saturating kernel (the shape of k_lk_fast: 1-wave workgroups, 128 VGPRs, 8 KB LDS, ~10^4 workgroups of ~90 us)
// owns the chip?  Varies the small kernel's workgroup size and LDS footprint.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(64, 4) void hog(int* out, int iters)
{
    __shared__ int lds[2048];   // 8 KB
    lds[threadIdx.x] = threadIdx.x;
    asm volatile("v_mov_b32 v127, 0" ::: "v127");   // forces a 128-VGPR allocation
    int a = threadIdx.x, b = blockIdx.x;
    iters *= 1 + ((blockIdx.x * 2654435761u) >> 29);   // 1..8 x: workgroups retire continuously, like tracked features
    for (int i = 0; i < iters; i++) { a = (a ^ b) + 0x1234567; b = (b ^ a) + 0x7654321; }
    out[(blockIdx.x * 64 + threadIdx.x) & 0xffff] = a + b + lds[(a & 63)];
}

template <int T, int LDSB>
__global__ __launch_bounds__(T) void small(int* out, int iters)
{
    __shared__ int lds[LDSB / 4 > 0 ? LDSB / 4 : 1];
    if (LDSB) lds[threadIdx.x % (LDSB / 4 > 0 ? LDSB / 4 : 1)] = threadIdx.x;
    int a = threadIdx.x;
    for (int i = 0; i < iters; i++) a = (a ^ i) + 0x1234567;
    out[(blockIdx.x * T + threadIdx.x) & 0xffff] = a + (LDSB ? lds[0] : 0);
}

template <int T, int LDSB>
static void run(const char* name, hipStream_t sa, hipStream_t sb, int* d, int hog_iters, int threads_total)
{
    hipEvent_t e0, e1, h0, h1;
    hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&h0); hipEventCreate(&h1);
    const int blocks = threads_total / T;
    float alone = 0, with = 0, hogms = 0;
    small<T, LDSB><<<blocks, T, 0, sb>>>(d, 200);
    hipDeviceSynchronize();
    hipEventRecord(e0, sb);
    small<T, LDSB><<<blocks, T, 0, sb>>>(d, 200);
    hipEventRecord(e1, sb);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&alone, e0, e1);
    // co-run: hog first, small kernel ~60 us later (host sleep by spinning on a timer)
    hipEventRecord(h0, sa);
    hog<<<10000, 64, 0, sa>>>(d, hog_iters);
    hipEventRecord(h1, sa);
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    do { clock_gettime(CLOCK_MONOTONIC, &t1); } while ((t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3 < 60.0);
    hipEventRecord(e0, sb);
    small<T, LDSB><<<blocks, T, 0, sb>>>(d, 200);
    hipEventRecord(e1, sb);
    hipDeviceSynchronize();
    hipEventElapsedTime(&with, e0, e1);
    hipEventElapsedTime(&hogms, h0, h1);
    printf("%-28s blocks %5d: alone %7.1f us   beside the hog %7.1f us   (hog %.1f us)\n", name, blocks, alone * 1e3,
           with * 1e3, hogms * 1e3);
}

int main()
{
    int* d; hipMalloc(&d, 65536 * 4);
    hipStream_t sa, sb;
    int least, greatest;
    hipDeviceGetStreamPriorityRange(&least, &greatest);
    hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
    hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, greatest);
    const int hog_iters = 1200;   // ~30 us x (1..8) per workgroup
    hog<<<10000, 64, 0, sa>>>(d, hog_iters);
    hipDeviceSynchronize();
    const int N = 256 * 1024;   // threads of the small kernel
    run<64, 0>("64 thr, no LDS", sa, sb, d, hog_iters, N);
    run<64, 4096>("64 thr, 4 KB LDS", sa, sb, d, hog_iters, N);
    run<64, 16384>("64 thr, 16 KB LDS", sa, sb, d, hog_iters, N);
    run<128, 0>("128 thr, no LDS", sa, sb, d, hog_iters, N);
    run<256, 0>("256 thr, no LDS", sa, sb, d, hog_iters, N);
    run<256, 16384>("256 thr, 16 KB LDS", sa, sb, d, hog_iters, N);
    run<256, 65536>("256 thr, 64 KB LDS", sa, sb, d, hog_iters, N);
    run<1024, 0>("1024 thr, no LDS", sa, sb, d, hog_iters, N);
    // the same with the small kernel on a NORMAL priority stream
    hipStream_t sc; hipStreamCreateWithFlags(&sc, hipStreamNonBlocking);
    run<64, 0>("64 thr, normal priority", sa, sc, d, hog_iters, N);
    run<256, 0>("256 thr, normal priority", sa, sc, d, hog_iters, N);
    return 0;
}
