// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU/LDS instruction kinds the cloud
// kernel is made of, on gfx950.  Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned u0 = threadIdx.x * 3 + 1, u1 = u0 + 5, u2 = u0 + 7, u3 = u0 + 11;
    for (int i = 0; i < iters; i++) {
        if constexpr (KIND == 0) { REP8(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %3, %3, %1, %2\n v_fma_f32 %4, %4, %1, %2\n v_fma_f32 %5, %5, %1, %2\n v_fma_f32 %6, %6, %1, %2\n v_fma_f32 %7, %7, %1, %2\n v_fma_f32 %8, %8, %1, %2\n v_fma_f32 %9, %9, %1, %2" : "+v"(a0) : "v"(a6), "v"(a7), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));) }
        if constexpr (KIND == 1) { REP8(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(i));) }
        if constexpr (KIND == 2) { REP8(asm volatile("v_cvt_f32_ubyte0 %0, %4\n v_cvt_f32_ubyte1 %1, %4\n v_cvt_f32_ubyte2 %2, %4\n v_cvt_f32_ubyte3 %3, %4\n v_cvt_f32_ubyte0 %0, %5\n v_cvt_f32_ubyte1 %1, %5\n v_cvt_f32_ubyte2 %2, %5\n v_cvt_f32_ubyte3 %3, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(u0), "v"(u1));) }
        if constexpr (KIND == 3) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3" : "+v"(*(double*)&a0), "+v"(*(double*)&a2) : "v"(*(double*)&a4), "v"(*(double*)&a6));) }
        if constexpr (KIND == 4) { REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if constexpr (KIND == 5) { REP8(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4) : "vcc");) }
        if constexpr (KIND == 6) { REP8(asm volatile("v_sub_u32_sdwa %0, %4, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_0\n v_sub_u32_sdwa %1, %4, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_2\n v_sub_u32_sdwa %2, %5, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_0\n v_sub_u32_sdwa %3, %5, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_2\n v_sub_u32_sdwa %0, %4, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_0\n v_sub_u32_sdwa %1, %4, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_2\n v_sub_u32_sdwa %2, %5, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_0\n v_sub_u32_sdwa %3, %5, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_2" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u0), "v"(u1));) }
        if constexpr (KIND == 7) { REP8(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));) }
        if constexpr (KIND == 8) { REP8(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1" : "+v"(*(unsigned long long*)&a0), "+v"(*(unsigned long long*)&a2) : "v"(u0), "v"(u1) : "vcc");) }
        if constexpr (KIND == 9) { REP8(asm volatile("v_exp_f32 %0, %0\n v_log_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_exp_f32 %3, %3\n v_log_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_exp_f32 %2, %2\n v_log_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if constexpr (KIND == 11) { REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]\n v_cndmask_b32_e64 %2, %2, %4, s[20:21]\n v_cndmask_b32_e64 %3, %3, %4, s[20:21]\n v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]\n v_cndmask_b32_e64 %2, %2, %4, s[20:21]\n v_cndmask_b32_e64 %3, %3, %4, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4) : "s20", "s21");) }
        if constexpr (KIND == 12) { REP8(asm volatile("v_cmp_gt_f32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_gt_f32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %4, vcc\n v_cmp_gt_f32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_gt_f32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4) : "vcc");) }
        if constexpr (KIND == 13) { REP8(asm volatile("v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %4, %5, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %4, %5, vcc\n v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %4, %5, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %4, %5, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5) : );) }
        if constexpr (KIND == 14) { REP8(asm volatile("v_pk_add_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2\n v_pk_add_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2\n v_pk_add_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2\n v_pk_add_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2" : "+v"(*(double*)&a0), "+v"(*(double*)&a2) : "v"(*(double*)&a4));) }
        if constexpr (KIND == 15) { REP8(asm volatile("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5\n v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5));) }
        if constexpr (KIND == 16) { REP8(asm volatile("v_mul_f32 %0, %0, %4 clamp\n v_floor_f32 %1, %1\n v_sub_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4 clamp\n v_floor_f32 %0, %0\n v_sub_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4 clamp\n v_floor_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));) }
        if constexpr (KIND == 17) { REP8(asm volatile("v_cmp_gt_f32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %1, %1, %5, vcc\n v_cndmask_b32 %2, %2, %5, vcc\n v_cndmask_b32 %3, %3, %5, vcc\n v_cndmask_b32 %1, %1, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5) : "vcc");) }
        if constexpr (KIND == 18) { REP8(asm volatile("v_cmp_gt_f32 s[20:21], %0, %4\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]\n v_cndmask_b32_e64 %2, %2, %4, s[20:21]\n v_cndmask_b32_e64 %3, %3, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %5, s[20:21]\n v_cndmask_b32_e64 %2, %2, %5, s[20:21]\n v_cndmask_b32_e64 %3, %3, %5, s[20:21]\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5) : "s20", "s21");) }
        if constexpr (KIND == 19) { REP8(asm volatile("v_fma_mix_f32 %0, %4, %5, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %4, %5, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %4, %5, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %4, %5, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %0, %4, %5, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %4, %5, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %4, %5, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %4, %5, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(u0), "v"(a5));) }
        if constexpr (KIND == 20) { REP8(asm volatile("v_cvt_f32_f16 %0, %4\n v_cvt_f32_f16_sdwa %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16 %2, %5\n v_cvt_f32_f16_sdwa %3, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16 %0, %4\n v_cvt_f32_f16_sdwa %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16 %2, %5\n v_cvt_f32_f16_sdwa %3, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(u0), "v"(u1));) }
        if constexpr (KIND == 10) { REP8(asm volatile("v_floor_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_f32_i32 %2, %2\n v_floor_f32 %3, %3\n v_and_b32 %0, %0, %4\n v_lshlrev_b32 %1, 3, %1\n v_max_f32 %2, %2, %4\n v_min_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));) }
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(u0 + u1 + u2 + u3) == 12345.678f) out[threadIdx.x] = a0;
}

template <int KIND> int run(const char* name, float* d, int n_per_iter) {
    const int iters = 2000, blocks = 256 * 8;   // 8 blocks of 256 per CU = 8 waves per SIMD
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    k<KIND><<<blocks, 256>>>(d, 10, 1.0f);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    k<KIND><<<blocks, 256>>>(d, iters, 1.0f);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    // per SIMD: waves = blocks*4/1024 = 8 ; wave-instructions per SIMD = 8 * iters * n_per_iter
    const double winstr_per_simd = 8.0 * iters * n_per_iter;
    const double cycles = ms * 1e-3 * 2.4e9;    // at nominal 2.4 GHz (actual clock may be lower)
    printf("%-28s %8.3f ms  %6.2f cycles/wave-instr/SIMD (at 2.4 GHz nominal)\n", name, ms, cycles / winstr_per_simd);
    return 0;
}

int main() {
    float* d; CHK(hipMalloc(&d, 4096));
    run<0>("v_fma_f32", d, 64); run<7>("v_mul/add_f32", d, 64); run<3>("v_pk_fma_f32", d, 64); run<1>("v_add_u32", d, 64);
    run<2>("v_cvt_f32_ubyteN", d, 64); run<5>("v_cndmask_b32", d, 64); run<6>("v_sub_u32_sdwa", d, 64); run<10>("floor/cvt/and/shl/min/max", d, 64);
    run<11>("v_cndmask_b32_e64 sgpr mask", d, 64); run<12>("v_cmp + v_cndmask vcc", d, 64); run<13>("v_cndmask vcc (no dep)", d, 64);
    run<17>("v_cmp vcc + 7 v_cndmask vcc", d, 64); run<18>("v_cmp sgpr + 7 v_cndmask e64", d, 64);
    run<19>("v_fma_mix_f32 (f16 src)", d, 64); run<20>("v_cvt_f32_f16 (+sdwa)", d, 64);
    run<14>("v_pk_add/mul_f32", d, 64); run<15>("v_fmac_f32", d, 64); run<16>("mul clamp/floor/sub", d, 64);
    run<4>("v_rcp_f32", d, 64); run<9>("v_exp/log/sqrt_f32", d, 64); run<8>("v_mad_u64_u32", d, 64);
    return 0;
}
