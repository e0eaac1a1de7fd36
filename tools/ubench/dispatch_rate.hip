// How fast does one launch get its workgroups onto the chip?  A kernel whose waves do nothing but wait `ticks` of the
// shader clock (s_memtime) and leave, with the register and LDS footprint of the 21x21 tracker (88 VGPRs, 5.4 KB: five
// waves per SIMD), launched as 20 016 waves in workgroups of 1, 2 and 4 waves: if waves per microsecond stop following
// slots / lifetime when the lifetime gets short, the dispatcher sets the pace, not the SIMDs.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NT>
__global__ __launch_bounds__(NT) void idle(int* out, unsigned ticks, int lds_words)
{
    extern __shared__ int lds[];
    asm volatile("v_mov_b32 v87, 0" ::: "v87");
    if (lds_words) lds[threadIdx.x] = 1;
    const unsigned long long t0 = __builtin_readcyclecounter();
    while ((unsigned long long)__builtin_readcyclecounter() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    if (out) out[0] = lds_words ? lds[threadIdx.x] : 0;
}

template <int NT>
static float run(int waves, unsigned ticks, int lds_bytes, int reps)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int wgs = waves / (NT / 64);
    idle<NT><<<wgs, NT, lds_bytes>>>(nullptr, ticks, lds_bytes / 4);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) idle<NT><<<wgs, NT, lds_bytes>>>(nullptr, ticks, lds_bytes / 4);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / reps;
}

int main()
{
    const int waves = 20016, reps = 5;
    // readcyclecounter = s_memtime: shader clock, about 2.1 GHz under load
    const unsigned ticks[] = {0, 24000, 48000, 96000, 120000};   // shader clock ticks (s_memtime)
    printf("waves %d, lifetime asked (us) -> launch us [waves/us] for 1 / 2 / 4 waves per workgroup (5.4 KB LDS per wave)\n", waves);
    for (unsigned t : ticks) {
        const float a = run<64>(waves, t, 5472, reps), b = run<128>(waves, t, 2 * 5472, reps), c = run<256>(waves, t, 4 * 5472, reps);
        printf("  %5.1f us: %8.1f [%6.1f]  %8.1f [%6.1f]  %8.1f [%6.1f]\n", t / 2100.0, a, waves / a, b, waves / b, c, waves / c);
    }
    return 0;
}
