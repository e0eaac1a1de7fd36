// VALU issue-rate microbenchmark: independent integer mads, W waves per SIMD (blocks of 64 threads).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int ILP>
__global__ __launch_bounds__(64) void k(int* out, int iters, int seed)
{
    int a[ILP];
#pragma unroll
    for (int i = 0; i < ILP; i++) a[i] = threadIdx.x * (i + 1) + seed;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) a[i] = (a[i] ^ a[(i + 1) % ILP]) + 0x1234567;   // v_xor + v_add (or one v_xad_u32): simple full-rate VALU
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
int main(int argc, char** argv)
{
    int* d; hipMalloc(&d, 64 * 65536 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096;
    for (int wps : {1, 2, 3, 4, 8}) {
        const int blocks = 256 * 4 * wps;   // wps waves on every SIMD
        k<8><<<blocks, 64>>>(d, iters, 1);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<8><<<blocks, 64>>>(d, iters, 2);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // VALU instrs per wave ~ iters * 8 * 2 (mul-add fused? count below from asm) ; report ns per (iter*ILP) per wave
        double per_simd_instr = (double)wps * iters * 8 * 2;
        printf("waves/SIMD %d: %.1f us, cycles per VALU instr per SIMD (assuming 2 instr per step, 2.4 GHz): %.2f\n", wps,
               ms * 1e3, ms * 1e-3 * 2.4e9 / per_simd_instr);
    }
    return 0;
}
