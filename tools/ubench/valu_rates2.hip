// valu_rates2.hip -- issue cost of gfx950 VALU instruction kinds in REAL shader cycles.
//
// Round-1's valu_rates.hip converted event milliseconds at a nominal 2.4 GHz and reported v_fma_f32 at 4.1 cycles per
// wave64 instruction per SIMD, while /opt/skills/guides/MI355X_MICROARCH.md:52-54 says "2 cycles (32 lanes/cycle x 2)".
// This version removes the clock assumption: every wave brackets its instruction stream with s_memtime (shader-clock
// ticks) AND s_memrealtime (constant 100 MHz), so
//   real clock          = d(s_memtime) / d(s_memrealtime) x 100 MHz          (what the chip actually ran at)
//   cycles / wave-instr = d(s_memtime) / (instructions of the wave x waves resident on its SIMD)
// with exactly W waves per SIMD on every CU (grid = 256 CUs x W workgroups of 4 waves), W in {1, 2, 4, 8}.  Every kind
// uses 8 independent accumulators, so at W >= 2 no wave waits on its own results; W = 1 shows the dependent-issue cost.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rates2.hip -o valu_rates2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <glob.h>
#include <fcntl.h>
#include <unistd.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x

// one test body = 8 instructions (string S) repeated 8 times = 64 instructions per loop iteration
#define BODY_SCALAR(S) REP8(asm volatile(S : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(u0), "v"(u1), "s"(s0) : "vcc", "s20", "s21");)
#define BODY_PACKED(S) REP8(asm volatile(S : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(q0), "v"(q1) : "vcc");)

struct Stamp { unsigned long long t0, t1, r0, r1; };

__device__ __forceinline__ unsigned long long memtime() { unsigned long long t; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }
__device__ __forceinline__ unsigned long long realtime() { unsigned long long t; asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }

#define KERNEL_SCALAR(NAME, S)                                                                                       \
    __global__ __launch_bounds__(256) void NAME(float* out, Stamp* st, int iters, float seed) {                      \
        float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float b0 = seed * 0.5f + 0.25f, b1 = seed * 0.25f + 1.0f, b2 = 1.0f - seed * 1e-3f, b3 = seed * 1e-4f;         \
        unsigned u0 = threadIdx.x * 3 + 1, u1 = threadIdx.x * 5 + 7;                                                \
        float s0 = seed * 0.75f;                                                                                     \
        const unsigned long long r0 = realtime(), t0 = memtime();                                                    \
        for (int i = 0; i < iters; i++) { BODY_SCALAR(S) }                                                           \
        const unsigned long long t1 = memtime(), r1 = realtime();                                                    \
        if ((threadIdx.x & 63) == 0) { Stamp s; s.t0 = t0; s.t1 = t1; s.r0 = r0; s.r1 = r1; st[blockIdx.x * 4 + (threadIdx.x >> 6)] = s; } \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[threadIdx.x] = a0;                             \
    }
#define KERNEL_PACKED(NAME, S)                                                                                       \
    __global__ __launch_bounds__(256) void NAME(float* out, Stamp* st, int iters, float seed) {                      \
        f2 p0 = {seed + threadIdx.x, seed}, p1 = p0 + 1.0f, p2 = p0 + 2.0f, p3 = p0 + 3.0f, p4 = p0 + 4.0f, p5 = p0 + 5.0f, p6 = p0 + 6.0f, p7 = p0 + 7.0f; \
        f2 q0 = {seed * 0.5f + 0.25f, 1.0f - seed * 1e-3f}, q1 = {seed * 1e-4f, seed * 2e-4f};                         \
        const unsigned long long r0 = realtime(), t0 = memtime();                                                    \
        for (int i = 0; i < iters; i++) { BODY_PACKED(S) }                                                           \
        const unsigned long long t1 = memtime(), r1 = realtime();                                                    \
        if ((threadIdx.x & 63) == 0) { Stamp s; s.t0 = t0; s.t1 = t1; s.r0 = r0; s.r1 = r1; st[blockIdx.x * 4 + (threadIdx.x >> 6)] = s; } \
        f2 z = p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7;                                                                \
        if (z.x + z.y == 12345.678f) out[threadIdx.x] = z.x;                                                        \
    }

// operands: %0..%7 accumulators, %8..%11 = b0..b3 (VGPR), %12, %13 = u0, u1 (VGPR), %14 = s0 (SGPR)
#define I8(op_fmt_a, ...) op_fmt_a
KERNEL_SCALAR(k_fma_3src, "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9")
KERNEL_SCALAR(k_fma_2src, "v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8")
KERNEL_SCALAR(k_fma_1src, "v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7")
KERNEL_SCALAR(k_fma_sgpr_const, "v_fma_f32 %0, %0, %14, 0.5\n v_fma_f32 %1, %1, %14, 0.5\n v_fma_f32 %2, %2, %14, 0.5\n v_fma_f32 %3, %3, %14, 0.5\n v_fma_f32 %4, %4, %14, 0.5\n v_fma_f32 %5, %5, %14, 0.5\n v_fma_f32 %6, %6, %14, 0.5\n v_fma_f32 %7, %7, %14, 0.5")
KERNEL_SCALAR(k_fmac, "v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9")
KERNEL_SCALAR(k_fma_mix, "v_fma_mix_f32 %0, %12, %8, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %12, %8, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %12, %8, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %12, %8, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %4, %13, %8, %4 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %13, %8, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %6, %13, %8, %6 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %13, %8, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]")
KERNEL_SCALAR(k_mul, "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8")
KERNEL_SCALAR(k_add, "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8")
KERNEL_SCALAR(k_mul_e64_2vgpr, "v_mul_f32_e64 %0, %8, %9\n v_mul_f32_e64 %1, %8, %9\n v_mul_f32_e64 %2, %8, %9\n v_mul_f32_e64 %3, %8, %9\n v_mul_f32_e64 %4, %8, %9\n v_mul_f32_e64 %5, %8, %9\n v_mul_f32_e64 %6, %8, %9\n v_mul_f32_e64 %7, %8, %9")
KERNEL_SCALAR(k_fma_mul_alt, "v_fma_f32 %0, %0, %8, %9\n v_mul_f32 %1, %1, %8\n v_fma_f32 %2, %2, %8, %9\n v_mul_f32 %3, %3, %8\n v_fma_f32 %4, %4, %8, %9\n v_mul_f32 %5, %5, %8\n v_fma_f32 %6, %6, %8, %9\n v_mul_f32 %7, %7, %8")
KERNEL_SCALAR(k_max_min, "v_max_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_min_f32 %7, %7, %8")
KERNEL_SCALAR(k_med3, "v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %8, %9")
KERNEL_SCALAR(k_mov, "v_mov_b32 %0, %8\n v_mov_b32 %1, %9\n v_mov_b32 %2, %8\n v_mov_b32 %3, %9\n v_mov_b32 %4, %8\n v_mov_b32 %5, %9\n v_mov_b32 %6, %8\n v_mov_b32 %7, %9")
KERNEL_SCALAR(k_and, "v_and_b32 %0, %0, %12\n v_and_b32 %1, %1, %12\n v_and_b32 %2, %2, %12\n v_and_b32 %3, %3, %12\n v_and_b32 %4, %4, %12\n v_and_b32 %5, %5, %12\n v_and_b32 %6, %6, %12\n v_and_b32 %7, %7, %12")
KERNEL_SCALAR(k_lshl, "v_lshlrev_b32 %0, 3, %0\n v_lshlrev_b32 %1, 3, %1\n v_lshlrev_b32 %2, 3, %2\n v_lshlrev_b32 %3, 3, %3\n v_lshlrev_b32 %4, 3, %4\n v_lshlrev_b32 %5, 3, %5\n v_lshlrev_b32 %6, 3, %6\n v_lshlrev_b32 %7, 3, %7")
KERNEL_SCALAR(k_lshl_or, "v_lshl_or_b32 %0, %0, 7, %12\n v_lshl_or_b32 %1, %1, 7, %12\n v_lshl_or_b32 %2, %2, 7, %12\n v_lshl_or_b32 %3, %3, 7, %12\n v_lshl_or_b32 %4, %4, 7, %12\n v_lshl_or_b32 %5, %5, 7, %12\n v_lshl_or_b32 %6, %6, 7, %12\n v_lshl_or_b32 %7, %7, 7, %12")
KERNEL_SCALAR(k_and_or, "v_and_or_b32 %0, %0, 63, %12\n v_and_or_b32 %1, %1, 63, %12\n v_and_or_b32 %2, %2, 63, %12\n v_and_or_b32 %3, %3, 63, %12\n v_and_or_b32 %4, %4, 63, %12\n v_and_or_b32 %5, %5, 63, %12\n v_and_or_b32 %6, %6, 63, %12\n v_and_or_b32 %7, %7, 63, %12")
KERNEL_SCALAR(k_bfe, "v_bfe_u32 %0, %0, 2, 5\n v_bfe_u32 %1, %1, 2, 5\n v_bfe_u32 %2, %2, 2, 5\n v_bfe_u32 %3, %3, 2, 5\n v_bfe_u32 %4, %4, 2, 5\n v_bfe_u32 %5, %5, 2, 5\n v_bfe_u32 %6, %6, 2, 5\n v_bfe_u32 %7, %7, 2, 5")
KERNEL_SCALAR(k_bfi, "v_bfi_b32 %0, %12, %0, %13\n v_bfi_b32 %1, %12, %1, %13\n v_bfi_b32 %2, %12, %2, %13\n v_bfi_b32 %3, %12, %3, %13\n v_bfi_b32 %4, %12, %4, %13\n v_bfi_b32 %5, %12, %5, %13\n v_bfi_b32 %6, %12, %6, %13\n v_bfi_b32 %7, %12, %7, %13")
KERNEL_SCALAR(k_add_u32, "v_add_u32 %0, %0, %12\n v_add_u32 %1, %1, %12\n v_add_u32 %2, %2, %12\n v_add_u32 %3, %3, %12\n v_add_u32 %4, %4, %12\n v_add_u32 %5, %5, %12\n v_add_u32 %6, %6, %12\n v_add_u32 %7, %7, %12")
KERNEL_SCALAR(k_lshl_add, "v_lshl_add_u32 %0, %0, 4, %12\n v_lshl_add_u32 %1, %1, 4, %12\n v_lshl_add_u32 %2, %2, 4, %12\n v_lshl_add_u32 %3, %3, 4, %12\n v_lshl_add_u32 %4, %4, 4, %12\n v_lshl_add_u32 %5, %5, 4, %12\n v_lshl_add_u32 %6, %6, 4, %12\n v_lshl_add_u32 %7, %7, 4, %12")
KERNEL_SCALAR(k_cvt_flr, "v_cvt_flr_i32_f32 %0, %8\n v_cvt_flr_i32_f32 %1, %9\n v_cvt_flr_i32_f32 %2, %8\n v_cvt_flr_i32_f32 %3, %9\n v_cvt_flr_i32_f32 %4, %8\n v_cvt_flr_i32_f32 %5, %9\n v_cvt_flr_i32_f32 %6, %8\n v_cvt_flr_i32_f32 %7, %9")
KERNEL_SCALAR(k_fract, "v_fract_f32 %0, %0\n v_fract_f32 %1, %1\n v_fract_f32 %2, %2\n v_fract_f32 %3, %3\n v_fract_f32 %4, %4\n v_fract_f32 %5, %5\n v_fract_f32 %6, %6\n v_fract_f32 %7, %7")
KERNEL_SCALAR(k_floor, "v_floor_f32 %0, %0\n v_floor_f32 %1, %1\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3\n v_floor_f32 %4, %4\n v_floor_f32 %5, %5\n v_floor_f32 %6, %6\n v_floor_f32 %7, %7")
KERNEL_SCALAR(k_cvt_f32_i32, "v_cvt_f32_i32 %0, %12\n v_cvt_f32_i32 %1, %13\n v_cvt_f32_i32 %2, %12\n v_cvt_f32_i32 %3, %13\n v_cvt_f32_i32 %4, %12\n v_cvt_f32_i32 %5, %13\n v_cvt_f32_i32 %6, %12\n v_cvt_f32_i32 %7, %13")
KERNEL_SCALAR(k_cmp_sgpr, "v_cmp_gt_f32 s[20:21], %0, %8\n v_cmp_gt_f32 s[20:21], %1, %8\n v_cmp_gt_f32 s[20:21], %2, %8\n v_cmp_gt_f32 s[20:21], %3, %8\n v_cmp_gt_f32 s[20:21], %4, %8\n v_cmp_gt_f32 s[20:21], %5, %8\n v_cmp_gt_f32 s[20:21], %6, %8\n v_cmp_gt_f32 s[20:21], %7, %8")
KERNEL_SCALAR(k_cndmask_sgpr, "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]")
KERNEL_SCALAR(k_rcp, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7")
KERNEL_SCALAR(k_exp, "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7")
KERNEL_SCALAR(k_log, "v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3\n v_log_f32 %4, %4\n v_log_f32 %5, %5\n v_log_f32 %6, %6\n v_log_f32 %7, %7")
KERNEL_SCALAR(k_sqrt, "v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7")
KERNEL_SCALAR(k_exp_fma_alt, "v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_exp_f32 %4, %4\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9")
KERNEL_SCALAR(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %12\n v_mul_lo_u32 %1, %1, %12\n v_mul_lo_u32 %2, %2, %12\n v_mul_lo_u32 %3, %3, %12\n v_mul_lo_u32 %4, %4, %12\n v_mul_lo_u32 %5, %5, %12\n v_mul_lo_u32 %6, %6, %12\n v_mul_lo_u32 %7, %7, %12")
KERNEL_SCALAR(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %12, %13\n v_mad_u32_u24 %1, %1, %12, %13\n v_mad_u32_u24 %2, %2, %12, %13\n v_mad_u32_u24 %3, %3, %12, %13\n v_mad_u32_u24 %4, %4, %12, %13\n v_mad_u32_u24 %5, %5, %12, %13\n v_mad_u32_u24 %6, %6, %12, %13\n v_mad_u32_u24 %7, %7, %12, %13")
KERNEL_SCALAR(k_cvt_f32_f16, "v_cvt_f32_f16 %0, %12\n v_cvt_f32_f16 %1, %13\n v_cvt_f32_f16 %2, %12\n v_cvt_f32_f16 %3, %13\n v_cvt_f32_f16 %4, %12\n v_cvt_f32_f16 %5, %13\n v_cvt_f32_f16 %6, %12\n v_cvt_f32_f16 %7, %13")
KERNEL_SCALAR(k_salu_between, "v_fma_f32 %0, %0, %8, %9\n s_nop 0\n v_fma_f32 %2, %2, %8, %9\n s_nop 0\n v_fma_f32 %4, %4, %8, %9\n s_nop 0\n v_fma_f32 %6, %6, %8, %9\n s_nop 0")
KERNEL_PACKED(k_pk_fma, "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9")
KERNEL_PACKED(k_pk_fma_2src, "v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8")
KERNEL_PACKED(k_pk_mul, "v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8")
KERNEL_PACKED(k_pk_add, "v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8")

// round 3: the kinds the basic-block census of the cloud kernel (tools/isa_profile.py) found without a measured cost, and a MIXED stream in
// the census's own proportions: is the per-kind pricing additive (does a stream of mixed kinds cost the sum of its kinds' costs)?
KERNEL_SCALAR(k_add_lshl, "v_add_lshl_u32 %0, %0, %12, 2\n v_add_lshl_u32 %1, %1, %12, 2\n v_add_lshl_u32 %2, %2, %12, 2\n v_add_lshl_u32 %3, %3, %12, 2\n v_add_lshl_u32 %4, %4, %12, 2\n v_add_lshl_u32 %5, %5, %12, 2\n v_add_lshl_u32 %6, %6, %12, 2\n v_add_lshl_u32 %7, %7, %12, 2")
KERNEL_SCALAR(k_or3, "v_or3_b32 %0, %0, %12, %13\n v_or3_b32 %1, %1, %12, %13\n v_or3_b32 %2, %2, %12, %13\n v_or3_b32 %3, %3, %12, %13\n v_or3_b32 %4, %4, %12, %13\n v_or3_b32 %5, %5, %12, %13\n v_or3_b32 %6, %6, %12, %13\n v_or3_b32 %7, %7, %12, %13")
KERNEL_SCALAR(k_readfirstlane, "v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %1\n v_readfirstlane_b32 s20, %2\n v_readfirstlane_b32 s21, %3\n v_readfirstlane_b32 s20, %4\n v_readfirstlane_b32 s21, %5\n v_readfirstlane_b32 s20, %6\n v_readfirstlane_b32 s21, %7")
KERNEL_SCALAR(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %12, %0\n v_mbcnt_hi_u32_b32 %1, %12, %1\n v_mbcnt_lo_u32_b32 %2, %12, %2\n v_mbcnt_hi_u32_b32 %3, %12, %3\n v_mbcnt_lo_u32_b32 %4, %12, %4\n v_mbcnt_hi_u32_b32 %5, %12, %5\n v_mbcnt_lo_u32_b32 %6, %12, %6\n v_mbcnt_hi_u32_b32 %7, %12, %7")
KERNEL_SCALAR(k_sub, "v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8")
KERNEL_SCALAR(k_fmamk, "v_fmamk_f32 %0, %0, 0x3f000000, %8\n v_fmamk_f32 %1, %1, 0x3f000000, %8\n v_fmamk_f32 %2, %2, 0x3f000000, %8\n v_fmamk_f32 %3, %3, 0x3f000000, %8\n v_fmamk_f32 %4, %4, 0x3f000000, %8\n v_fmamk_f32 %5, %5, 0x3f000000, %8\n v_fmamk_f32 %6, %6, 0x3f000000, %8\n v_fmamk_f32 %7, %7, 0x3f000000, %8")
// 16 instructions in the proportions of the cloud kernel's executed mix (census, C3): 3 mul, 2 fma, 2 fmac, add, sub, 2 fma_mix, mov, and, cvt_flr, fract, lshl
KERNEL_SCALAR(k_mix16, "v_mul_f32 %0, %0, %8\n v_fma_f32 %1, %1, %8, %9\n v_add_f32 %2, %2, %8\n v_fmac_f32 %3, %8, %9\n v_fma_mix_f32 %4, %12, %8, %4 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_mov_b32 %5, %8\n v_cvt_flr_i32_f32 %6, %9\n v_fract_f32 %7, %7\n"
                       " v_mul_f32 %0, %0, %9\n v_fmac_f32 %1, %8, %9\n v_sub_f32 %2, %2, %9\n v_fma_mix_f32 %3, %13, %8, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_and_b32 %5, %5, %12\n v_lshlrev_b32 %6, 3, %6\n v_mul_f32 %7, %7, %8\n v_fma_f32 %4, %4, %8, %9")
// the same 16 with one transcendental in place of the mov (the kernel runs 1 transcendental per ~21 VALU instructions)
KERNEL_SCALAR(k_mix16_trans, "v_mul_f32 %0, %0, %8\n v_fma_f32 %1, %1, %8, %9\n v_add_f32 %2, %2, %8\n v_fmac_f32 %3, %8, %9\n v_fma_mix_f32 %4, %12, %8, %4 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_rcp_f32 %5, %5\n v_cvt_flr_i32_f32 %6, %9\n v_fract_f32 %7, %7\n"
                             " v_mul_f32 %0, %0, %9\n v_fmac_f32 %1, %8, %9\n v_sub_f32 %2, %2, %9\n v_fma_mix_f32 %3, %13, %8, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_and_b32 %5, %5, %12\n v_lshlrev_b32 %6, 3, %6\n v_mul_f32 %7, %7, %8\n v_fma_f32 %4, %4, %8, %9")

// round 6 (the packed-fp16 filter experiment, profiles/r06/fp16_filter_ab.txt): what the two instructions it swapped in cost to issue
KERNEL_SCALAR(k_pk_fma_f16, "v_pk_fma_f16 %0, %12, %13, %0\n v_pk_fma_f16 %1, %12, %13, %1\n v_pk_fma_f16 %2, %12, %13, %2\n v_pk_fma_f16 %3, %12, %13, %3\n v_pk_fma_f16 %4, %12, %13, %4\n v_pk_fma_f16 %5, %12, %13, %5\n v_pk_fma_f16 %6, %12, %13, %6\n v_pk_fma_f16 %7, %12, %13, %7")
KERNEL_SCALAR(k_cvt_pk_f16, "v_cvt_pk_f16_f32 %0, %8, %9\n v_cvt_pk_f16_f32 %1, %9, %8\n v_cvt_pk_f16_f32 %2, %8, %9\n v_cvt_pk_f16_f32 %3, %9, %8\n v_cvt_pk_f16_f32 %4, %8, %9\n v_cvt_pk_f16_f32 %5, %9, %8\n v_cvt_pk_f16_f32 %6, %8, %9\n v_cvt_pk_f16_f32 %7, %9, %8")

typedef void (*kern_t)(float*, Stamp*, int, float);
struct Test { const char* name; kern_t k; int valu_per_8; };   // VALU instructions among the 8 of one body line

static int run(const Test& t, float* d_out, Stamp* d_st, std::vector<Stamp>& h, int cus) {
    const int iters = 1500;
    printf("%-22s", t.name);
    for (int W : {1, 2, 4, 8}) {
        const int blocks = cus * W;
        t.k<<<blocks, 256>>>(d_out, d_st, 8, 1.0f);                   // warm-up (code fetch, clocks)
        CHK(hipDeviceSynchronize());
        t.k<<<blocks, 256>>>(d_out, d_st, iters, 1.0f);
        CHK(hipDeviceSynchronize());
        CHK(hipMemcpy(h.data(), d_st, sizeof(Stamp) * blocks * 4, hipMemcpyDeviceToHost));
        double cyc = 0.0, mhz = 0.0;
        for (int i = 0; i < blocks * 4; i++) {
            cyc += (double)(h[i].t1 - h[i].t0);
            mhz += (double)(h[i].t1 - h[i].t0) / (double)(h[i].r1 - h[i].r0) * 100.0;
        }
        cyc /= blocks * 4; mhz /= blocks * 4;
        const double per_instr = cyc / ((double)iters * 8.0 * t.valu_per_8 * W);   // W waves share each SIMD
        printf("  W=%d %5.2f cyc @%4.0f MHz", W, per_instr, mhz);
    }
    printf("\n");
    return 0;
}

// ---- round 4: DIFFERENTIAL issue cost ("a full-rate kind must read ~2.0, not 2.3" if loop / launch overhead is what inflates it).
// Every kind is launched at 8 waves per SIMD with N and 3N loop iterations, timed with HIP events; the cost per instruction is the SLOPE
//   (t(3N) - t(N)) / (2N x 64 instructions x 8 waves)   x   the shader clock sampled from the amdgpu hwmon node while the 3N launch ran
// -- launch ramp, prologue, the stamps and the tail drop out of the difference, and the unit (milliseconds x sampled MHz) is exactly the one
// bench.py's roofline uses for the cycles a SIMD had.  What remains inside the slope is the loop's own 3 scalar instructions per 64 VALU.
struct Sclk {
    std::string path; std::atomic<bool> stop{false}; std::vector<double> mhz; std::thread th;
    void start() { stop = false; mhz.clear(); th = std::thread([this] {
        int fd = open(path.c_str(), O_RDONLY); if (fd < 0) return;
        char b[64];
        while (!stop) { lseek(fd, 0, SEEK_SET); ssize_t n = read(fd, b, 63); if (n > 0) { b[n] = 0; double v = atof(b); if (v > 0) mhz.push_back(v / 1e6); } usleep(250); }
        close(fd); }); }
    double finish() { stop = true; if (th.joinable()) th.join(); double a = 0; for (double v : mhz) a += v; return mhz.empty() ? 0.0 : a / mhz.size(); }
};
static int run_diff(const Test& t, float* d_out, Stamp* d_st, int cus, Sclk& clk, FILE* js, bool first) {
    const int W = 8, blocks = cus * W, N = 12000;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float ms[2] = {0, 0}; double mhz = 0.0;
    t.k<<<blocks, 256>>>(d_out, d_st, 64, 1.0f); CHK(hipDeviceSynchronize());
    for (int pass = 0; pass < 2; pass++) {
        const int iters = pass == 0 ? N : 3 * N;
        if (pass == 1) clk.start();
        CHK(hipEventRecord(e0)); t.k<<<blocks, 256>>>(d_out, d_st, iters, 1.0f); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        if (pass == 1) mhz = clk.finish();
        CHK(hipEventElapsedTime(&ms[pass], e0, e1));
    }
    const double instr = 2.0 * N * 8.0 * t.valu_per_8 * W;                       // extra instructions per SIMD in the longer launch
    const double ns = (ms[1] - ms[0]) * 1e6 / instr, whole = ms[1] * 1e6 / (3.0 * N * 8.0 * t.valu_per_8 * W);
    printf("%-28s slope %6.4f ns/instr  x %6.1f MHz = %5.3f cycles   (whole launch / instructions: %5.3f cycles; t(N) %.3f ms, t(3N) %.3f ms)\n",
           t.name, ns, mhz, ns * mhz * 1e-3, whole * mhz * 1e-3, ms[0], ms[1]);
    if (js) fprintf(js, "%s\n  \"%s\": {\"cycles\": %.4f, \"ns_per_instr\": %.5f, \"sclk_mhz\": %.1f, \"cycles_whole_launch\": %.4f}", first ? "" : ",", t.name, ns * mhz * 1e-3, ns, mhz, whole * mhz * 1e-3);
    CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
    return 0;
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const bool diff = argc > 1 && !strcmp(argv[1], "diff");
    const char* only = getenv("VALU_RATES_ONLY");              // substring filter on the kind names (a short run for one question)
    if (!diff)
    printf("# %s, %d CUs, clockRate %d kHz.  cycles = s_memtime ticks per wave64 VALU instruction per SIMD with W waves resident per SIMD;\n"
           "# MHz = s_memtime / s_memrealtime (100 MHz) over the same interval = the clock the chip actually sustained\n", prop.gcnArchName, cus, prop.clockRate);
    float* d_out; Stamp* d_st;
    CHK(hipMalloc(&d_out, 4096));
    CHK(hipMalloc(&d_st, sizeof(Stamp) * cus * 8 * 4));
    std::vector<Stamp> h((size_t)cus * 8 * 4);
    const Test tests[] = {
        {"v_fma_f32 3 vgpr src", k_fma_3src, 8}, {"v_fma_f32 2 vgpr src", k_fma_2src, 8}, {"v_fma_f32 1 vgpr src", k_fma_1src, 8},
        {"v_fma_f32 sgpr+const", k_fma_sgpr_const, 8}, {"v_fmac_f32", k_fmac, 8}, {"v_fma_mix_f32", k_fma_mix, 8},
        {"v_mul_f32", k_mul, 8}, {"v_add_f32", k_add, 8}, {"v_mul_f32_e64 2 vgpr", k_mul_e64_2vgpr, 8}, {"fma/mul alternating", k_fma_mul_alt, 8},
        {"v_max/min_f32", k_max_min, 8}, {"v_med3_f32", k_med3, 8}, {"v_mov_b32", k_mov, 8}, {"v_and_b32", k_and, 8}, {"v_lshlrev_b32", k_lshl, 8},
        {"v_lshl_or_b32", k_lshl_or, 8}, {"v_and_or_b32", k_and_or, 8}, {"v_bfe_u32", k_bfe, 8}, {"v_bfi_b32", k_bfi, 8}, {"v_add_u32", k_add_u32, 8},
        {"v_lshl_add_u32", k_lshl_add, 8}, {"v_cvt_flr_i32_f32", k_cvt_flr, 8}, {"v_fract_f32", k_fract, 8}, {"v_floor_f32", k_floor, 8},
        {"v_cvt_f32_i32", k_cvt_f32_i32, 8}, {"v_cvt_f32_f16", k_cvt_f32_f16, 8}, {"v_cmp_gt_f32 -> sgpr", k_cmp_sgpr, 8}, {"v_cndmask_b32 sgpr", k_cndmask_sgpr, 8},
        {"v_rcp_f32", k_rcp, 8}, {"v_exp_f32", k_exp, 8}, {"v_log_f32", k_log, 8}, {"v_sqrt_f32", k_sqrt, 8}, {"1 exp : 3 fma", k_exp_fma_alt, 8},
        {"v_mul_lo_u32", k_mul_lo_u32, 8}, {"v_mad_u32_u24", k_mad_u32_u24, 8}, {"fma + s_nop alternating", k_salu_between, 4},
        {"v_add_lshl_u32", k_add_lshl, 8}, {"v_or3_b32", k_or3, 8}, {"v_readfirstlane_b32", k_readfirstlane, 8}, {"v_mbcnt_lo/hi", k_mbcnt, 8}, {"v_sub_f32", k_sub, 8},
        {"v_fmamk_f32", k_fmamk, 8}, {"mix16 (census proportions)", k_mix16, 16}, {"mix16 with 1 rcp", k_mix16_trans, 16},
        {"v_pk_fma_f32 3 src", k_pk_fma, 8}, {"v_pk_fma_f32 2 src", k_pk_fma_2src, 8}, {"v_pk_mul_f32", k_pk_mul, 8}, {"v_pk_add_f32", k_pk_add, 8},
        {"v_pk_fma_f16", k_pk_fma_f16, 8}, {"v_cvt_pk_f16_f32", k_cvt_pk_f16, 8},
    };
    if (diff) {                                                // valu_rates2 diff [out.json]
        Sclk clk;
        glob_t g; if (glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input", 0, nullptr, &g) == 0 && g.gl_pathc > 0) clk.path = g.gl_pathv[0];
        printf("# %s, %d CUs: differential issue cost at 8 waves per SIMD, clock sampled from %s\n", prop.gcnArchName, cus, clk.path.empty() ? "(no hwmon node: MHz = 0)" : clk.path.c_str());
        FILE* js = argc > 2 ? fopen(argv[2], "w") : nullptr;
        if (js) fprintf(js, "{");
        bool first = true;
        for (const Test& t : tests) { if (only && !strstr(only, t.name)) continue; if (run_diff(t, d_out, d_st, cus, clk, js, first)) return 1; first = false; }
        if (js) { fprintf(js, "\n}\n"); fclose(js); }
        return 0;
    }
    for (const Test& t : tests) { if (only && !strstr(only, t.name)) continue; if (run(t, d_out, d_st, h, cus)) return 1; }
    // the peak the spec sheet quotes: 157.3 TFLOP/s = 256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz.  64 FLOP/clk/SIMD is reached by
    // a wave64 v_fma_f32 every 2 cycles OR a wave64 v_pk_fma_f32 (2 FMAs per lane) every 4 cycles: compare with the rows above.
    return 0;
}
