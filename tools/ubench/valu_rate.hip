// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction per SIMD for the instruction classes
// the tracker kernel (k_lk_fast.hip) is made of, at 1 / 2 / 4 / 8 waves per SIMD.
//
// Every measured loop body is ONE inline-asm block of exactly 16 instructions of one kind over 8 independent
// registers (two rounds), so the instruction count is known by construction and can be checked in the
// disassembly (`llvm-objdump -d`: each loop is 16 x <op> + s_sub + s_cmp + s_cbranch).  Workgroups are 256
// threads = one wave on each of the CU's four SIMDs; `wps` workgroups per CU give wps waves per SIMD.
//
// s_memtime counts shader cycles, s_memrealtime a constant 100 MHz: their ratio is the clock the wave really saw.
// A launch of w workgroups per CU does not always keep w of them resident together, so the wave's own span is
// compared with the launch's span to see how many waves shared its SIMD (the xN in the output).
//
// build: hipcc --offload-arch=gfx950 -O2 -o valu_rate valu_rate.hip ; run: ./valu_rate > profiles/rNN_valu_rate.txt
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum Op {
    OP_ADD_U32, OP_XAD_U32, OP_MUL_I24, OP_MAD_I24, OP_DOT2_I16, OP_PERM, OP_ALIGNBYTE, OP_PK_ADD_U16, OP_PK_MAD_U16,
    OP_PK_MUL_LO_U16, OP_MUL_LO_U32, OP_FMA_F32, OP_PK_FMA_F32, OP_ADD_DPP, OP_CNDMASK, OP_LSHRREV, OP_READLANE,
    OP_S_ADD, OP_DS_READ_B32, OP_ASHRREV, OP_CNDMASK_E64, OP_DOT2C, OP_MOV, OP_ADD3, OP_LSHL_OR, OP_RNDNE, OP_CVT_I32_F32,
    OP_CVT_F64_I32, OP_ADD_F64, OP_MUL_F64, OP_CVT_F32_F64, OP_DS_READ_B64, OP_DS_READ2_B32, OP_AND_OR, OP_MAX_I32, OP_CMP_CND_VCC, OP_CMP_CND_SGPR, OP_CMP_VCC, OP_CMP_SGPR, OP_ALIGNBIT, OP_BFE, OP_MIN_U32, OP_SUB_U32, OP_AND, OP_OR, OP_LSHL_ADD, OP_MED3, OP_COUNT
};
static const char* kNames[OP_COUNT] = {
    "v_add_u32", "v_xad_u32", "v_mul_i32_i24", "v_mad_i32_i24", "v_dot2_i32_i16", "v_perm_b32", "v_alignbyte_b32",
    "v_pk_add_u16", "v_pk_mad_u16", "v_pk_mul_lo_u16", "v_mul_lo_u32", "v_fma_f32", "v_pk_fma_f32",
    "v_add_u32 row_shr:1 (DPP)", "v_cndmask_b32", "v_lshrrev_b32", "v_readlane_b32 (-> SGPR)", "s_add_u32 (SALU)",
    "ds_read_b32 (+ lgkmcnt wait per 16)", "v_ashrrev_i32", "v_cndmask_b32_e64 (SGPR-pair mask)", "v_dot2c_i32_i16 (VOP2)",
    "v_mov_b32", "v_add3_u32", "v_lshl_or_b32", "v_rndne_f32", "v_cvt_i32_f32", "v_cvt_f64_i32", "v_add_f64", "v_mul_f64",
    "v_cvt_f32_f64", "ds_read_b64 (+ wait per 16)", "ds_read2_b32 (+ wait per 16)", "v_and_or_b32", "v_max_i32",
    "v_cmp_gt_i32 vcc + v_cndmask vcc (per instruction)", "v_cmp_gt_i32 s[..] + v_cndmask s[..] (per instruction)",
    "v_cmp_gt_i32_e32 (-> vcc)", "v_cmp_gt_i32_e64 (-> SGPR pair)", "v_alignbit_b32", "v_bfe_u32", "v_min_u32", "v_sub_u32",
    "v_and_b32", "v_or_b32", "v_lshl_add_u32", "v_med3_i32",
};

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned long long* ticks, int iters, unsigned seed)
{
    __shared__ unsigned lds[2048];
    unsigned r0 = threadIdx.x + seed, r1 = r0 * 3 + 1, r2 = r0 * 5 + 2, r3 = r0 * 7 + 3, r4 = r0 * 11 + 4, r5 = r0 * 13 + 5,
             r6 = r0 * 17 + 6, r7 = r0 * 19 + 7;
    unsigned c = seed | 0x01010101u;
    unsigned long long p0 = r0, p1 = r1, p2 = r2, p3 = r3;   // 64-bit registers for the packed-f32 case
    lds[threadIdx.x] = r0; lds[threadIdx.x + 256] = r1; lds[threadIdx.x + 512] = r2; lds[threadIdx.x + 768] = r3;
    __syncthreads();
    unsigned addr = (threadIdx.x & 63) * 4;
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#define BODY3(op)                                                                                                       \
    asm volatile(op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op      \
                    " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n" op " %0, %0, %8\n" op " %1, %1, %8\n" op      \
                    " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op      \
                    " %7, %7, %8\n"                                                                                     \
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)                       \
                 : "v"(c))
#define BODY4(op)                                                                                                       \
    asm volatile(op " %0, %0, %8, %0\n" op " %1, %1, %8, %1\n" op " %2, %2, %8, %2\n" op " %3, %3, %8, %3\n" op         \
                    " %4, %4, %8, %4\n" op " %5, %5, %8, %5\n" op " %6, %6, %8, %6\n" op " %7, %7, %8, %7\n" op         \
                    " %0, %0, %8, %0\n" op " %1, %1, %8, %1\n" op " %2, %2, %8, %2\n" op " %3, %3, %8, %3\n" op         \
                    " %4, %4, %8, %4\n" op " %5, %5, %8, %5\n" op " %6, %6, %8, %6\n" op " %7, %7, %8, %7\n"            \
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)                       \
                 : "v"(c))
        if constexpr (OP == OP_ADD_U32) BODY3("v_add_u32");
        else if constexpr (OP == OP_XAD_U32) BODY4("v_xad_u32");
        else if constexpr (OP == OP_MUL_I24) BODY3("v_mul_i32_i24");
        else if constexpr (OP == OP_MAD_I24) BODY4("v_mad_i32_i24");
        else if constexpr (OP == OP_DOT2_I16) BODY4("v_dot2_i32_i16");
        else if constexpr (OP == OP_PERM) BODY4("v_perm_b32");
        else if constexpr (OP == OP_ALIGNBYTE) BODY4("v_alignbyte_b32");
        else if constexpr (OP == OP_PK_ADD_U16) BODY3("v_pk_add_u16");
        else if constexpr (OP == OP_PK_MAD_U16) BODY4("v_pk_mad_u16");
        else if constexpr (OP == OP_PK_MUL_LO_U16) BODY3("v_pk_mul_lo_u16");
        else if constexpr (OP == OP_MUL_LO_U32) BODY3("v_mul_lo_u32");
        else if constexpr (OP == OP_FMA_F32) BODY4("v_fma_f32");
        else if constexpr (OP == OP_CNDMASK) BODY3("v_cndmask_b32");
        else if constexpr (OP == OP_LSHRREV) BODY3("v_lshrrev_b32");
        else if constexpr (OP == OP_ASHRREV) BODY3("v_ashrrev_i32");
        else if constexpr (OP == OP_DOT2C) BODY3("v_dot2c_i32_i16");
        else if constexpr (OP == OP_ADD3) BODY4("v_add3_u32");
        else if constexpr (OP == OP_LSHL_OR) BODY4("v_lshl_or_b32");
        else if constexpr (OP == OP_AND_OR) BODY4("v_and_or_b32");
        else if constexpr (OP == OP_MAX_I32) BODY3("v_max_i32");
        else if constexpr (OP == OP_ALIGNBIT) BODY4("v_alignbit_b32");
        else if constexpr (OP == OP_BFE) BODY4("v_bfe_u32");
        else if constexpr (OP == OP_MIN_U32) BODY3("v_min_u32");
        else if constexpr (OP == OP_SUB_U32) BODY3("v_sub_u32");
        else if constexpr (OP == OP_AND) BODY3("v_and_b32");
        else if constexpr (OP == OP_OR) BODY3("v_or_b32");
        else if constexpr (OP == OP_LSHL_ADD) BODY4("v_lshl_add_u32");
        else if constexpr (OP == OP_MED3) BODY4("v_med3_i32");
        else if constexpr (OP == OP_CMP_CND_VCC) {
#define P(i) "v_cmp_gt_i32 vcc, %" #i ", %8\nv_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
            asm volatile(REP8(P)
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                         : "v"(c) : "vcc");
#undef P
        } else if constexpr (OP == OP_CMP_CND_SGPR) {
            unsigned long long m;
#define P(i) "v_cmp_gt_i32 %9, %" #i ", %8\nv_cndmask_b32 %" #i ", %" #i ", %8, %9\n"
            asm volatile(REP8(P)
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                         : "v"(c), "s"(m = 0));
#undef P
        } else if constexpr (OP == OP_CMP_VCC) {
#define P(i) "v_cmp_gt_i32 vcc, %" #i ", %8\n"
            asm volatile(REP8(P) REP8(P) : : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7), "v"(c) : "vcc");
#undef P
        } else if constexpr (OP == OP_CMP_SGPR) {
            unsigned long long m;
#define P(i) "v_cmp_gt_i32 %0, %" "1" ", %9\n"
            asm volatile("v_cmp_gt_i32 %0, %1, %9\nv_cmp_gt_i32 %0, %2, %9\nv_cmp_gt_i32 %0, %3, %9\nv_cmp_gt_i32 %0, %4, %9\n"
                         "v_cmp_gt_i32 %0, %5, %9\nv_cmp_gt_i32 %0, %6, %9\nv_cmp_gt_i32 %0, %7, %9\nv_cmp_gt_i32 %0, %8, %9\n"
                         "v_cmp_gt_i32 %0, %1, %9\nv_cmp_gt_i32 %0, %2, %9\nv_cmp_gt_i32 %0, %3, %9\nv_cmp_gt_i32 %0, %4, %9\n"
                         "v_cmp_gt_i32 %0, %5, %9\nv_cmp_gt_i32 %0, %6, %9\nv_cmp_gt_i32 %0, %7, %9\nv_cmp_gt_i32 %0, %8, %9\n"
                         : "=s"(m) : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7), "v"(c));
#undef P
            r0 ^= (unsigned)m & 1u;
        }
        else if constexpr (OP == OP_MOV) {
#define D(i) "v_mov_b32 %" #i ", %8\n"
            asm volatile(REP8(D) REP8(D)
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7)
                         : "v"(c));
#undef D
        } else if constexpr (OP == OP_RNDNE || OP == OP_CVT_I32_F32) {
#define D(i) OPSTR " %" #i ", %" #i "\n"
            if constexpr (OP == OP_RNDNE) {
#define OPSTR "v_rndne_f32"
                asm volatile(REP8(D) REP8(D) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));
#undef OPSTR
            } else {
#define OPSTR "v_cvt_i32_f32"
                asm volatile(REP8(D) REP8(D) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));
#undef OPSTR
            }
#undef D
        } else if constexpr (OP == OP_CNDMASK_E64) {
            unsigned long long m = 0x5555aaaa3333ccccull ^ c;
            asm volatile(
                "v_cndmask_b32 %0, %0, %8, %9\nv_cndmask_b32 %1, %1, %8, %9\nv_cndmask_b32 %2, %2, %8, %9\nv_cndmask_b32 %3, %3, %8, %9\n"
                "v_cndmask_b32 %4, %4, %8, %9\nv_cndmask_b32 %5, %5, %8, %9\nv_cndmask_b32 %6, %6, %8, %9\nv_cndmask_b32 %7, %7, %8, %9\n"
                "v_cndmask_b32 %0, %0, %8, %9\nv_cndmask_b32 %1, %1, %8, %9\nv_cndmask_b32 %2, %2, %8, %9\nv_cndmask_b32 %3, %3, %8, %9\n"
                "v_cndmask_b32 %4, %4, %8, %9\nv_cndmask_b32 %5, %5, %8, %9\nv_cndmask_b32 %6, %6, %8, %9\nv_cndmask_b32 %7, %7, %8, %9\n"
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                : "v"(c), "s"(m));
        } else if constexpr (OP == OP_CVT_F64_I32 || OP == OP_ADD_F64 || OP == OP_MUL_F64 || OP == OP_CVT_F32_F64) {
            if constexpr (OP == OP_CVT_F64_I32)
                asm volatile("v_cvt_f64_i32 %0, %4\nv_cvt_f64_i32 %1, %5\nv_cvt_f64_i32 %2, %6\nv_cvt_f64_i32 %3, %7\n"
                             "v_cvt_f64_i32 %0, %4\nv_cvt_f64_i32 %1, %5\nv_cvt_f64_i32 %2, %6\nv_cvt_f64_i32 %3, %7\n"
                             "v_cvt_f64_i32 %0, %4\nv_cvt_f64_i32 %1, %5\nv_cvt_f64_i32 %2, %6\nv_cvt_f64_i32 %3, %7\n"
                             "v_cvt_f64_i32 %0, %4\nv_cvt_f64_i32 %1, %5\nv_cvt_f64_i32 %2, %6\nv_cvt_f64_i32 %3, %7\n"
                             : "=v"(p0), "=v"(p1), "=v"(p2), "=v"(p3) : "v"(r0), "v"(r1), "v"(r2), "v"(r3));
            else if constexpr (OP == OP_CVT_F32_F64)
                asm volatile("v_cvt_f32_f64 %0, %4\nv_cvt_f32_f64 %1, %5\nv_cvt_f32_f64 %2, %6\nv_cvt_f32_f64 %3, %7\n"
                             "v_cvt_f32_f64 %0, %4\nv_cvt_f32_f64 %1, %5\nv_cvt_f32_f64 %2, %6\nv_cvt_f32_f64 %3, %7\n"
                             "v_cvt_f32_f64 %0, %4\nv_cvt_f32_f64 %1, %5\nv_cvt_f32_f64 %2, %6\nv_cvt_f32_f64 %3, %7\n"
                             "v_cvt_f32_f64 %0, %4\nv_cvt_f32_f64 %1, %5\nv_cvt_f32_f64 %2, %6\nv_cvt_f32_f64 %3, %7\n"
                             : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3));
            else if constexpr (OP == OP_ADD_F64)
                asm volatile("v_add_f64 %0, %0, %4\nv_add_f64 %1, %1, %4\nv_add_f64 %2, %2, %4\nv_add_f64 %3, %3, %4\n"
                             "v_add_f64 %0, %0, %4\nv_add_f64 %1, %1, %4\nv_add_f64 %2, %2, %4\nv_add_f64 %3, %3, %4\n"
                             "v_add_f64 %0, %0, %4\nv_add_f64 %1, %1, %4\nv_add_f64 %2, %2, %4\nv_add_f64 %3, %3, %4\n"
                             "v_add_f64 %0, %0, %4\nv_add_f64 %1, %1, %4\nv_add_f64 %2, %2, %4\nv_add_f64 %3, %3, %4\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"((unsigned long long)c << 20));
            else
                asm volatile("v_mul_f64 %0, %0, %4\nv_mul_f64 %1, %1, %4\nv_mul_f64 %2, %2, %4\nv_mul_f64 %3, %3, %4\n"
                             "v_mul_f64 %0, %0, %4\nv_mul_f64 %1, %1, %4\nv_mul_f64 %2, %2, %4\nv_mul_f64 %3, %3, %4\n"
                             "v_mul_f64 %0, %0, %4\nv_mul_f64 %1, %1, %4\nv_mul_f64 %2, %2, %4\nv_mul_f64 %3, %3, %4\n"
                             "v_mul_f64 %0, %0, %4\nv_mul_f64 %1, %1, %4\nv_mul_f64 %2, %2, %4\nv_mul_f64 %3, %3, %4\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"((unsigned long long)c << 20));
        } else if constexpr (OP == OP_DS_READ_B64) {
            asm volatile("ds_read_b64 %0, %4 offset:0\nds_read_b64 %1, %4 offset:512\nds_read_b64 %2, %4 offset:1024\nds_read_b64 %3, %4 offset:1536\n"
                         "ds_read_b64 %0, %4 offset:2048\nds_read_b64 %1, %4 offset:2560\nds_read_b64 %2, %4 offset:3072\nds_read_b64 %3, %4 offset:3584\n"
                         "ds_read_b64 %0, %4 offset:0\nds_read_b64 %1, %4 offset:512\nds_read_b64 %2, %4 offset:1024\nds_read_b64 %3, %4 offset:1536\n"
                         "ds_read_b64 %0, %4 offset:2048\nds_read_b64 %1, %4 offset:2560\nds_read_b64 %2, %4 offset:3072\nds_read_b64 %3, %4 offset:3584\n"
                         "s_waitcnt lgkmcnt(0)\n"
                         : "=v"(p0), "=v"(p1), "=v"(p2), "=v"(p3) : "v"(addr * 2) : "memory");
        } else if constexpr (OP == OP_DS_READ2_B32) {
            asm volatile("ds_read2_b32 %0, %4 offset0:0 offset1:65\nds_read2_b32 %1, %4 offset0:130 offset1:195\nds_read2_b32 %2, %4 offset0:2 offset1:67\nds_read2_b32 %3, %4 offset0:132 offset1:197\n"
                         "ds_read2_b32 %0, %4 offset0:4 offset1:69\nds_read2_b32 %1, %4 offset0:134 offset1:199\nds_read2_b32 %2, %4 offset0:6 offset1:71\nds_read2_b32 %3, %4 offset0:136 offset1:201\n"
                         "ds_read2_b32 %0, %4 offset0:0 offset1:65\nds_read2_b32 %1, %4 offset0:130 offset1:195\nds_read2_b32 %2, %4 offset0:2 offset1:67\nds_read2_b32 %3, %4 offset0:132 offset1:197\n"
                         "ds_read2_b32 %0, %4 offset0:4 offset1:69\nds_read2_b32 %1, %4 offset0:134 offset1:199\nds_read2_b32 %2, %4 offset0:6 offset1:71\nds_read2_b32 %3, %4 offset0:136 offset1:201\n"
                         "s_waitcnt lgkmcnt(0)\n"
                         : "=v"(p0), "=v"(p1), "=v"(p2), "=v"(p3) : "v"(addr) : "memory");
        }
        else if constexpr (OP == OP_PK_FMA_F32) {
            asm volatile(
                "v_pk_fma_f32 %0, %0, %4, %0\nv_pk_fma_f32 %1, %1, %4, %1\nv_pk_fma_f32 %2, %2, %4, %2\nv_pk_fma_f32 %3, %3, %4, %3\n"
                "v_pk_fma_f32 %0, %0, %4, %0\nv_pk_fma_f32 %1, %1, %4, %1\nv_pk_fma_f32 %2, %2, %4, %2\nv_pk_fma_f32 %3, %3, %4, %3\n"
                "v_pk_fma_f32 %0, %0, %4, %0\nv_pk_fma_f32 %1, %1, %4, %1\nv_pk_fma_f32 %2, %2, %4, %2\nv_pk_fma_f32 %3, %3, %4, %3\n"
                "v_pk_fma_f32 %0, %0, %4, %0\nv_pk_fma_f32 %1, %1, %4, %1\nv_pk_fma_f32 %2, %2, %4, %2\nv_pk_fma_f32 %3, %3, %4, %3\n"
                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)
                : "v"(p0 ^ p1));
        } else if constexpr (OP == OP_ADD_DPP) {
#define D(i) "v_add_u32_dpp %" #i ", %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
            asm volatile(REP8(D) REP8(D)
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));
#undef D
        } else if constexpr (OP == OP_READLANE) {
            unsigned s0, s1, s2, s3, s4, s5, s6, s7;
            asm volatile(
                "v_readlane_b32 %0, %8, 7\nv_readlane_b32 %1, %9, 15\nv_readlane_b32 %2, %10, 23\nv_readlane_b32 %3, %11, 31\n"
                "v_readlane_b32 %4, %12, 39\nv_readlane_b32 %5, %13, 47\nv_readlane_b32 %6, %14, 55\nv_readlane_b32 %7, %15, 63\n"
                "v_readlane_b32 %0, %8, 8\nv_readlane_b32 %1, %9, 16\nv_readlane_b32 %2, %10, 24\nv_readlane_b32 %3, %11, 32\n"
                "v_readlane_b32 %4, %12, 40\nv_readlane_b32 %5, %13, 48\nv_readlane_b32 %6, %14, 56\nv_readlane_b32 %7, %15, 62\n"
                : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3), "=s"(s4), "=s"(s5), "=s"(s6), "=s"(s7)
                : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7));
            c += s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7;   // a handful of SALU + 1 VALU, not counted
        } else if constexpr (OP == OP_S_ADD) {
            unsigned s0 = c, s1 = c + 1, s2 = c + 2, s3 = c + 3;
            asm volatile(
                "s_add_u32 %0, %0, %1\ns_add_u32 %1, %1, %2\ns_add_u32 %2, %2, %3\ns_add_u32 %3, %3, %0\n"
                "s_add_u32 %0, %0, %1\ns_add_u32 %1, %1, %2\ns_add_u32 %2, %2, %3\ns_add_u32 %3, %3, %0\n"
                "s_add_u32 %0, %0, %1\ns_add_u32 %1, %1, %2\ns_add_u32 %2, %2, %3\ns_add_u32 %3, %3, %0\n"
                "s_add_u32 %0, %0, %1\ns_add_u32 %1, %1, %2\ns_add_u32 %2, %2, %3\ns_add_u32 %3, %3, %0\n"
                : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)
                :
                : "scc");
            c = s0 + s1 + s2 + s3;
        } else if constexpr (OP == OP_DS_READ_B32) {
#define D(i) "ds_read_b32 %" #i ", %8 offset:" #i "*256\n"
            asm volatile(REP8(D) REP8(D) "s_waitcnt lgkmcnt(0)\n"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7)
                         : "v"(addr)
                         : "memory");
#undef D
            addr = (addr + (r0 & 4)) & 1023;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    unsigned s = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ c ^ (unsigned)(p0 ^ p1 ^ p2 ^ p3) ^ addr;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        ticks[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = t1 - t0;
        ticks[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = rt1 - rt0;
    }
}

template <int OP>
static void run(unsigned* d_out, unsigned long long* d_ticks, int n_cu, double clock_hz)
{
    const int iters = 8192;
    printf("%-36s", kNames[OP]);
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = n_cu * wps;
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, d_ticks, iters, 1u);   // warm-up
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, d_ticks, iters, 2u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> t(blocks * 8);
        hipMemcpy(t.data(), d_ticks, sizeof(unsigned long long) * t.size(), hipMemcpyDeviceToHost);
        std::vector<double> cyc, us;
        for (int i = 0; i < blocks * 4; i++) { cyc.push_back((double)t[2 * i]); us.push_back((double)t[2 * i + 1] * 1e-2); }
        std::sort(cyc.begin(), cyc.end());
        std::sort(us.begin(), us.end());
        const double med_cyc = cyc[cyc.size() / 2], med_us = us[us.size() / 2];   // s_memtime cycles / s_memrealtime (100 MHz)
        const double instr_per_simd = (double)wps * iters * 16.0;
        // wall: HIP-event time of the whole launch at the nominal clock.  wave: the median wave's own span in shader
        // cycles x (waves per SIMD that overlapped it = its span / the launch's span, at most wps) -- i.e. SIMD cycles
        // per instruction while that wave ran; clk = the shader clock the wave saw (cycles / real time)
        const double cyc_wall = ms * 1e-3 * clock_hz / instr_per_simd;
        const double clk_ghz = med_cyc / med_us * 1e-3;
        const double conc = std::min((double)wps, wps * med_us / (ms * 1e3));   // waves per SIMD resident together
        const double cyc_simd = med_cyc / (iters * 16.0) / conc;
        printf("  w%d: %5.2f | %5.2f @%.2fGHz x%.1f", wps, cyc_wall, cyc_simd, clk_ghz, conc);
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    printf("\n");
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    const double clock_hz = prop.clockRate * 1e3;
    printf("# %s, %d CUs, nominal shader clock %.0f MHz\n", prop.name, n_cu, clock_hz / 1e6);
    printf("# cycles per wave64 instruction per SIMD with w workgroups of 256 threads per CU (= w waves per SIMD asked for);\n"
           "# 16 instructions per loop body over 8 independent registers (4 for 64-bit operands).  Per column:\n"
           "#   A | B @clk xN :  A = HIP-event time of the launch x nominal clock / instructions per SIMD\n"
           "#                    B = median wave's s_memtime cycles per own instruction / N, N = waves per SIMD resident together\n"
           "#                    (wave span / launch span x w, capped at w), clk = s_memtime cycles / s_memrealtime\n");
    unsigned* d_out;
    unsigned long long* d_ticks;
    hipMalloc(&d_out, sizeof(unsigned) * 256 * n_cu * 8);
    hipMalloc(&d_ticks, sizeof(unsigned long long) * 8 * n_cu * 8);
    run<OP_ADD_U32>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_XAD_U32>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_LSHRREV>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_ASHRREV>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_CNDMASK>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_MUL_I24>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_MAD_I24>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_DOT2_I16>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_PERM>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_ALIGNBYTE>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_PK_ADD_U16>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_PK_MAD_U16>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_PK_MUL_LO_U16>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_MUL_LO_U32>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_FMA_F32>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_PK_FMA_F32>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_ADD_DPP>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_READLANE>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_S_ADD>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_DS_READ_B32>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_DS_READ_B64>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_DS_READ2_B32>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_CNDMASK_E64>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_DOT2C>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_MOV>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_ADD3>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_LSHL_OR>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_AND_OR>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_MAX_I32>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_CMP_CND_VCC>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_CMP_CND_SGPR>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_CMP_VCC>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_CMP_SGPR>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_ALIGNBIT>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_BFE>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_MIN_U32>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_SUB_U32>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_AND>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_OR>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_LSHL_ADD>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_MED3>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_RNDNE>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_CVT_I32_F32>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_CVT_F64_I32>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_ADD_F64>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_MUL_F64>(d_out, d_ticks, n_cu, clock_hz);
    run<OP_CVT_F32_F64>(d_out, d_ticks, n_cu, clock_hz);
    return 0;
}
