// valu_ops.hip -- GPU-box micro-benchmark: issue cost, in cycles per
// wave64 instruction per SIMD, of the VALU instruction kinds the sequential voice kernels are made of.
// Each kind: one asm block of 8 independent instructions, unrolled 8x, at 1 and 8 waves per SIMD
// (at 8 neither dependency latency nor occupancy limits the rate).
// Build: hipcc --offload-arch=gfx950 -O3 valu_ops.hip -o valu_ops
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(64) k_ops(float *out, unsigned iters, float a, float b) {
    float x[8];
    f2 p[8];
    const f2 bb = {b, b * 0.5f};
#pragma unroll
    for (int k = 0; k < 8; k++) { x[k] = a + (float)(threadIdx.x + k); p[k] = f2{x[k], x[k] + 1.0f}; }
    asm volatile("s_mov_b64 vcc, 0x5555\n\ts_mov_b64 s[20:21], 0x3333\n\ts_mov_b64 s[22:23], 0" ::: "vcc", "s20", "s21", "s22", "s23");
    for (unsigned i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if constexpr (KIND == 0) asm volatile("v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %8\n\tv_add_f32 %2, %2, %8\n\tv_add_f32 %3, %3, %8\n\tv_add_f32 %4, %4, %8\n\tv_add_f32 %5, %5, %8\n\tv_add_f32 %6, %6, %8\n\tv_add_f32 %7, %7, %8" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 1) asm volatile("v_mul_f32 %0, %0, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %7, %7, %8" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 2) asm volatile("v_sub_f32 %0, %0, %8\n\tv_sub_f32 %1, %1, %8\n\tv_sub_f32 %2, %2, %8\n\tv_sub_f32 %3, %3, %8\n\tv_sub_f32 %4, %4, %8\n\tv_sub_f32 %5, %5, %8\n\tv_sub_f32 %6, %6, %8\n\tv_sub_f32 %7, %7, %8" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 3) asm volatile("v_add_f32_e64 %0, -%0, %8\n\tv_add_f32_e64 %1, -%1, %8\n\tv_add_f32_e64 %2, -%2, %8\n\tv_add_f32_e64 %3, -%3, %8\n\tv_add_f32_e64 %4, -%4, %8\n\tv_add_f32_e64 %5, -%5, %8\n\tv_add_f32_e64 %6, -%6, %8\n\tv_add_f32_e64 %7, -%7, %8" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 4) asm volatile("v_add_f32 %0, s20, %0\n\tv_add_f32 %1, s20, %1\n\tv_add_f32 %2, s20, %2\n\tv_add_f32 %3, s20, %3\n\tv_add_f32 %4, s20, %4\n\tv_add_f32 %5, s20, %5\n\tv_add_f32 %6, s20, %6\n\tv_add_f32 %7, s20, %7" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 5) asm volatile("v_add_f32 %0, 0x36800000, %0\n\tv_add_f32 %1, 0x36800000, %1\n\tv_add_f32 %2, 0x36800000, %2\n\tv_add_f32 %3, 0x36800000, %3\n\tv_add_f32 %4, 0x36800000, %4\n\tv_add_f32 %5, 0x36800000, %5\n\tv_add_f32 %6, 0x36800000, %6\n\tv_add_f32 %7, 0x36800000, %7" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 6) asm volatile("v_min_f32 %0, %0, %8\n\tv_min_f32 %1, %1, %8\n\tv_min_f32 %2, %2, %8\n\tv_min_f32 %3, %3, %8\n\tv_min_f32 %4, %4, %8\n\tv_min_f32 %5, %5, %8\n\tv_min_f32 %6, %6, %8\n\tv_min_f32 %7, %7, %8" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 7) asm volatile("v_max_f32 %0, %0, %8\n\tv_max_f32 %1, %1, %8\n\tv_max_f32 %2, %2, %8\n\tv_max_f32 %3, %3, %8\n\tv_max_f32 %4, %4, %8\n\tv_max_f32 %5, %5, %8\n\tv_max_f32 %6, %6, %8\n\tv_max_f32 %7, %7, %8" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 8) asm volatile("v_fma_f32 %0, %0, %8, %8\n\tv_fma_f32 %1, %1, %8, %8\n\tv_fma_f32 %2, %2, %8, %8\n\tv_fma_f32 %3, %3, %8, %8\n\tv_fma_f32 %4, %4, %8, %8\n\tv_fma_f32 %5, %5, %8, %8\n\tv_fma_f32 %6, %6, %8, %8\n\tv_fma_f32 %7, %7, %8, %8" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 9) asm volatile("v_med3_f32 %0, %0, %8, %8\n\tv_med3_f32 %1, %1, %8, %8\n\tv_med3_f32 %2, %2, %8, %8\n\tv_med3_f32 %3, %3, %8, %8\n\tv_med3_f32 %4, %4, %8, %8\n\tv_med3_f32 %5, %5, %8, %8\n\tv_med3_f32 %6, %6, %8, %8\n\tv_med3_f32 %7, %7, %8, %8" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 10) asm volatile("v_pk_add_f32 %0, %0, %8\n\tv_pk_add_f32 %1, %1, %8\n\tv_pk_add_f32 %2, %2, %8\n\tv_pk_add_f32 %3, %3, %8\n\tv_pk_add_f32 %4, %4, %8\n\tv_pk_add_f32 %5, %5, %8\n\tv_pk_add_f32 %6, %6, %8\n\tv_pk_add_f32 %7, %7, %8" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(bb));
            if constexpr (KIND == 11) asm volatile("v_pk_mul_f32 %0, %0, %8\n\tv_pk_mul_f32 %1, %1, %8\n\tv_pk_mul_f32 %2, %2, %8\n\tv_pk_mul_f32 %3, %3, %8\n\tv_pk_mul_f32 %4, %4, %8\n\tv_pk_mul_f32 %5, %5, %8\n\tv_pk_mul_f32 %6, %6, %8\n\tv_pk_mul_f32 %7, %7, %8" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(bb));
            if constexpr (KIND == 12) asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n\tv_pk_fma_f32 %1, %1, %8, %8\n\tv_pk_fma_f32 %2, %2, %8, %8\n\tv_pk_fma_f32 %3, %3, %8, %8\n\tv_pk_fma_f32 %4, %4, %8, %8\n\tv_pk_fma_f32 %5, %5, %8, %8\n\tv_pk_fma_f32 %6, %6, %8, %8\n\tv_pk_fma_f32 %7, %7, %8, %8" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(bb));
            if constexpr (KIND == 13) asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_cndmask_b32 %2, %2, %8, vcc\n\tv_cndmask_b32 %3, %3, %8, vcc\n\tv_cndmask_b32 %4, %4, %8, vcc\n\tv_cndmask_b32 %5, %5, %8, vcc\n\tv_cndmask_b32 %6, %6, %8, vcc\n\tv_cndmask_b32 %7, %7, %8, vcc" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 14) asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n\tv_cndmask_b32_e64 %1, %1, %8, s[20:21]\n\tv_cndmask_b32_e64 %2, %2, %8, s[20:21]\n\tv_cndmask_b32_e64 %3, %3, %8, s[20:21]\n\tv_cndmask_b32_e64 %4, %4, %8, s[20:21]\n\tv_cndmask_b32_e64 %5, %5, %8, s[20:21]\n\tv_cndmask_b32_e64 %6, %6, %8, s[20:21]\n\tv_cndmask_b32_e64 %7, %7, %8, s[20:21]" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 15) asm volatile("v_cmp_lt_f32 vcc, %0, %8\n\tv_cmp_lt_f32 vcc, %1, %8\n\tv_cmp_lt_f32 vcc, %2, %8\n\tv_cmp_lt_f32 vcc, %3, %8\n\tv_cmp_lt_f32 vcc, %4, %8\n\tv_cmp_lt_f32 vcc, %5, %8\n\tv_cmp_lt_f32 vcc, %6, %8\n\tv_cmp_lt_f32 vcc, %7, %8" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 16) asm volatile("v_cmp_lt_u32 vcc, %0, %8\n\tv_cmp_lt_u32 vcc, %1, %8\n\tv_cmp_lt_u32 vcc, %2, %8\n\tv_cmp_lt_u32 vcc, %3, %8\n\tv_cmp_lt_u32 vcc, %4, %8\n\tv_cmp_lt_u32 vcc, %5, %8\n\tv_cmp_lt_u32 vcc, %6, %8\n\tv_cmp_lt_u32 vcc, %7, %8" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 17) asm volatile("v_cmp_lt_u32_e64 s[22:23], %0, %8\n\tv_cmp_lt_u32_e64 s[22:23], %1, %8\n\tv_cmp_lt_u32_e64 s[22:23], %2, %8\n\tv_cmp_lt_u32_e64 s[22:23], %3, %8\n\tv_cmp_lt_u32_e64 s[22:23], %4, %8\n\tv_cmp_lt_u32_e64 s[22:23], %5, %8\n\tv_cmp_lt_u32_e64 s[22:23], %6, %8\n\tv_cmp_lt_u32_e64 s[22:23], %7, %8" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 18) asm volatile("v_cmp_lt_u32 vcc, %0, %8\n\tv_cndmask_b32 %0, %0, %8, vcc\n\tv_cmp_lt_u32 vcc, %1, %8\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_cmp_lt_u32 vcc, %2, %8\n\tv_cndmask_b32 %2, %2, %8, vcc\n\tv_cmp_lt_u32 vcc, %3, %8\n\tv_cndmask_b32 %3, %3, %8, vcc\n\tv_cmp_lt_u32 vcc, %4, %8\n\tv_cndmask_b32 %4, %4, %8, vcc\n\tv_cmp_lt_u32 vcc, %5, %8\n\tv_cndmask_b32 %5, %5, %8, vcc\n\tv_cmp_lt_u32 vcc, %6, %8\n\tv_cndmask_b32 %6, %6, %8, vcc\n\tv_cmp_lt_u32 vcc, %7, %8\n\tv_cndmask_b32 %7, %7, %8, vcc" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 19) asm volatile("v_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\tv_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 20) asm volatile("v_sub_co_u32 %0, vcc, %0, %8\n\tv_sub_co_u32 %1, vcc, %1, %8\n\tv_sub_co_u32 %2, vcc, %2, %8\n\tv_sub_co_u32 %3, vcc, %3, %8\n\tv_sub_co_u32 %4, vcc, %4, %8\n\tv_sub_co_u32 %5, vcc, %5, %8\n\tv_sub_co_u32 %6, vcc, %6, %8\n\tv_sub_co_u32 %7, vcc, %7, %8" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 21) asm volatile("v_and_b32 %0, %0, %8\n\tv_and_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_and_b32 %3, %3, %8\n\tv_and_b32 %4, %4, %8\n\tv_and_b32 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_and_b32 %7, %7, %8" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 22) asm volatile("v_xor_b32 %0, %0, %8\n\tv_xor_b32 %1, %1, %8\n\tv_xor_b32 %2, %2, %8\n\tv_xor_b32 %3, %3, %8\n\tv_xor_b32 %4, %4, %8\n\tv_xor_b32 %5, %5, %8\n\tv_xor_b32 %6, %6, %8\n\tv_xor_b32 %7, %7, %8" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 23) asm volatile("v_bfi_b32 %0, %8, %0, %8\n\tv_bfi_b32 %1, %8, %1, %8\n\tv_bfi_b32 %2, %8, %2, %8\n\tv_bfi_b32 %3, %8, %3, %8\n\tv_bfi_b32 %4, %8, %4, %8\n\tv_bfi_b32 %5, %8, %5, %8\n\tv_bfi_b32 %6, %8, %6, %8\n\tv_bfi_b32 %7, %8, %7, %8" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 24) asm volatile("v_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %4, 1, %4\n\tv_lshrrev_b32 %5, 1, %5\n\tv_lshrrev_b32 %6, 1, %6\n\tv_lshrrev_b32 %7, 1, %7" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 25) asm volatile("v_ashrrev_i32 %0, 31, %0\n\tv_ashrrev_i32 %1, 31, %1\n\tv_ashrrev_i32 %2, 31, %2\n\tv_ashrrev_i32 %3, 31, %3\n\tv_ashrrev_i32 %4, 31, %4\n\tv_ashrrev_i32 %5, 31, %5\n\tv_ashrrev_i32 %6, 31, %6\n\tv_ashrrev_i32 %7, 31, %7" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 26) asm volatile("v_mov_b32 %0, %8\n\tv_mov_b32 %1, %8\n\tv_mov_b32 %2, %8\n\tv_mov_b32 %3, %8\n\tv_mov_b32 %4, %8\n\tv_mov_b32 %5, %8\n\tv_mov_b32 %6, %8\n\tv_mov_b32 %7, %8" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 27) asm volatile("v_cvt_f32_u32 %0, %0\n\tv_cvt_f32_u32 %1, %1\n\tv_cvt_f32_u32 %2, %2\n\tv_cvt_f32_u32 %3, %3\n\tv_cvt_f32_u32 %4, %4\n\tv_cvt_f32_u32 %5, %5\n\tv_cvt_f32_u32 %6, %6\n\tv_cvt_f32_u32 %7, %7" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 28) asm volatile("v_lshl_add_u32 %0, %0, 1, %8\n\tv_lshl_add_u32 %1, %1, 1, %8\n\tv_lshl_add_u32 %2, %2, 1, %8\n\tv_lshl_add_u32 %3, %3, 1, %8\n\tv_lshl_add_u32 %4, %4, 1, %8\n\tv_lshl_add_u32 %5, %5, 1, %8\n\tv_lshl_add_u32 %6, %6, 1, %8\n\tv_lshl_add_u32 %7, %7, 1, %8" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 29) asm volatile("v_and_or_b32 %0, %0, %8, %8\n\tv_and_or_b32 %1, %1, %8, %8\n\tv_and_or_b32 %2, %2, %8, %8\n\tv_and_or_b32 %3, %3, %8, %8\n\tv_and_or_b32 %4, %4, %8, %8\n\tv_and_or_b32 %5, %5, %8, %8\n\tv_and_or_b32 %6, %6, %8, %8\n\tv_and_or_b32 %7, %7, %8, %8" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 30) asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 31) asm volatile("v_lshlrev_b64 %0, 1, %0\n\tv_lshlrev_b64 %1, 1, %1\n\tv_lshlrev_b64 %2, 1, %2\n\tv_lshlrev_b64 %3, 1, %3\n\tv_lshlrev_b64 %4, 1, %4\n\tv_lshlrev_b64 %5, 1, %5\n\tv_lshlrev_b64 %6, 1, %6\n\tv_lshlrev_b64 %7, 1, %7" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(bb));
            if constexpr (KIND == 32) asm volatile("v_lshrrev_b64 %0, 1, %0\n\tv_lshrrev_b64 %1, 1, %1\n\tv_lshrrev_b64 %2, 1, %2\n\tv_lshrrev_b64 %3, 1, %3\n\tv_lshrrev_b64 %4, 1, %4\n\tv_lshrrev_b64 %5, 1, %5\n\tv_lshrrev_b64 %6, 1, %6\n\tv_lshrrev_b64 %7, 1, %7" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(bb));
            if constexpr (KIND == 33) asm volatile("v_lshl_add_u64 %0, %0, 0, %8\n\tv_lshl_add_u64 %1, %1, 0, %8\n\tv_lshl_add_u64 %2, %2, 0, %8\n\tv_lshl_add_u64 %3, %3, 0, %8\n\tv_lshl_add_u64 %4, %4, 0, %8\n\tv_lshl_add_u64 %5, %5, 0, %8\n\tv_lshl_add_u64 %6, %6, 0, %8\n\tv_lshl_add_u64 %7, %7, 0, %8" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(bb));
            if constexpr (KIND == 34) asm volatile("v_cmp_gt_u64 vcc, %0, %8\n\tv_cmp_gt_u64 vcc, %1, %8\n\tv_cmp_gt_u64 vcc, %2, %8\n\tv_cmp_gt_u64 vcc, %3, %8\n\tv_cmp_gt_u64 vcc, %4, %8\n\tv_cmp_gt_u64 vcc, %5, %8\n\tv_cmp_gt_u64 vcc, %6, %8\n\tv_cmp_gt_u64 vcc, %7, %8" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(bb) : "vcc");
            if constexpr (KIND == 35) asm volatile("v_alignbit_b32 %0, %0, %8, 9\n\tv_alignbit_b32 %1, %1, %8, 9\n\tv_alignbit_b32 %2, %2, %8, 9\n\tv_alignbit_b32 %3, %3, %8, 9\n\tv_alignbit_b32 %4, %4, %8, 9\n\tv_alignbit_b32 %5, %5, %8, 9\n\tv_alignbit_b32 %6, %6, %8, 9\n\tv_alignbit_b32 %7, %7, %8, 9" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 36) asm volatile("v_ffbh_u32 %0, %0\n\tv_ffbh_u32 %1, %1\n\tv_ffbh_u32 %2, %2\n\tv_ffbh_u32 %3, %3\n\tv_ffbh_u32 %4, %4\n\tv_ffbh_u32 %5, %5\n\tv_ffbh_u32 %6, %6\n\tv_ffbh_u32 %7, %7" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 37) asm volatile("v_min3_u32 %0, %0, %8, 64\n\tv_min3_u32 %1, %1, %8, 64\n\tv_min3_u32 %2, %2, %8, 64\n\tv_min3_u32 %3, %3, %8, 64\n\tv_min3_u32 %4, %4, %8, 64\n\tv_min3_u32 %5, %5, %8, 64\n\tv_min3_u32 %6, %6, %8, 64\n\tv_min3_u32 %7, %7, %8, 64" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 38) asm volatile("v_add_co_u32 %0, vcc, %0, %8\n\tv_addc_co_u32 %1, vcc, %1, %8, vcc\n\tv_add_co_u32 %2, vcc, %2, %8\n\tv_addc_co_u32 %3, vcc, %3, %8, vcc\n\tv_add_co_u32 %4, vcc, %4, %8\n\tv_addc_co_u32 %5, vcc, %5, %8, vcc\n\tv_add_co_u32 %6, vcc, %6, %8\n\tv_addc_co_u32 %7, vcc, %7, %8, vcc" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b) : "vcc");
            if constexpr (KIND == 39) asm volatile("v_add_u32_e64 %0, %0, 32 clamp\n\tv_add_u32_e64 %1, %1, 32 clamp\n\tv_add_u32_e64 %2, %2, 32 clamp\n\tv_add_u32_e64 %3, %3, 32 clamp\n\tv_add_u32_e64 %4, %4, 32 clamp\n\tv_add_u32_e64 %5, %5, 32 clamp\n\tv_add_u32_e64 %6, %6, 32 clamp\n\tv_add_u32_e64 %7, %7, 32 clamp" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 40) asm volatile("v_mul_lo_u32 %0, %0, %8\n\tv_mul_lo_u32 %1, %1, %8\n\tv_mul_lo_u32 %2, %2, %8\n\tv_mul_lo_u32 %3, %3, %8\n\tv_mul_lo_u32 %4, %4, %8\n\tv_mul_lo_u32 %5, %5, %8\n\tv_mul_lo_u32 %6, %6, %8\n\tv_mul_lo_u32 %7, %7, %8" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(b));
            if constexpr (KIND == 41) asm volatile("v_mul_f64 %0, %0, %8\n\tv_mul_f64 %1, %1, %8\n\tv_mul_f64 %2, %2, %8\n\tv_mul_f64 %3, %3, %8\n\tv_mul_f64 %4, %4, %8\n\tv_mul_f64 %5, %5, %8\n\tv_mul_f64 %6, %6, %8\n\tv_mul_f64 %7, %7, %8" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(bb));
            if constexpr (KIND == 42) asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(x[0]) : "v"(b));
            if constexpr (KIND == 43) asm volatile("v_xor_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1" : "+v"(x[0]) : "v"(b));
            if constexpr (KIND == 44) asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1" : "+v"(x[0]) : "v"(b));
            if constexpr (KIND == 45) asm volatile("v_lshlrev_b64 %0, 1, %0\n\tv_lshlrev_b64 %0, 1, %0\n\tv_lshlrev_b64 %0, 1, %0\n\tv_lshlrev_b64 %0, 1, %0\n\tv_lshlrev_b64 %0, 1, %0\n\tv_lshlrev_b64 %0, 1, %0\n\tv_lshlrev_b64 %0, 1, %0\n\tv_lshlrev_b64 %0, 1, %0" : "+v"(p[0]) : "v"(bb));
            if constexpr (KIND == 46) asm volatile("v_alignbit_b32 %0, %0, %1, 9\n\tv_alignbit_b32 %0, %0, %1, 9\n\tv_alignbit_b32 %0, %0, %1, 9\n\tv_alignbit_b32 %0, %0, %1, 9\n\tv_alignbit_b32 %0, %0, %1, 9\n\tv_alignbit_b32 %0, %0, %1, 9\n\tv_alignbit_b32 %0, %0, %1, 9\n\tv_alignbit_b32 %0, %0, %1, 9" : "+v"(x[0]) : "v"(b));
            if constexpr (KIND == 47) asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2" : "+v"(x[0]), "+v"(x[1]) : "v"(b));
            if constexpr (KIND == 48) asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(x[0]) : "v"(b));
            if constexpr (KIND == 49) asm volatile("v_mul_f32 %0, %0, %1\n\tv_sub_f32 %0, %1, %0\n\tv_xor_b32 %0, %0, %1\n\tv_lshrrev_b32 %0, 1, %0\n\tv_mul_f32 %0, %0, %1\n\tv_sub_f32 %0, %1, %0\n\tv_xor_b32 %0, %0, %1\n\tv_lshrrev_b32 %0, 1, %0" : "+v"(x[0]) : "v"(b));
            if constexpr (KIND == 50) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[0]) : "v"(b) : "vcc");
            if constexpr (KIND == 51) asm volatile("v_mul_f64 %0, %0, %1\n\tv_add_f64 %0, %0, %1\n\tv_mul_f64 %0, %0, %1\n\tv_add_f64 %0, %0, %1\n\tv_mul_f64 %0, %0, %1\n\tv_add_f64 %0, %0, %1\n\tv_mul_f64 %0, %0, %1\n\tv_add_f64 %0, %0, %1" : "+v"(p[0]) : "v"(bb));
            if constexpr (KIND == 52) asm volatile("v_mul_f32 %0, 0x3e99999a, %0\n\tv_add_f32 %0, 0xb6800000, %0\n\tv_mul_f32 %0, 0x3e99999a, %0\n\tv_add_f32 %0, 0xb6800000, %0\n\tv_mul_f32 %0, 0x3e99999a, %0\n\tv_add_f32 %0, 0xb6800000, %0\n\tv_mul_f32 %0, 0x3e99999a, %0\n\tv_add_f32 %0, 0xb6800000, %0" : "+v"(x[0]) : "v"(b));
            if constexpr (KIND == 53) asm volatile("v_fma_f64 %0, %0, %8, %8\n\tv_fma_f64 %1, %1, %8, %8\n\tv_fma_f64 %2, %2, %8, %8\n\tv_fma_f64 %3, %3, %8, %8\n\tv_fma_f64 %4, %4, %8, %8\n\tv_fma_f64 %5, %5, %8, %8\n\tv_fma_f64 %6, %6, %8, %8\n\tv_fma_f64 %7, %7, %8, %8" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(bb), "v"(b));
            if constexpr (KIND == 54) asm volatile("v_rndne_f64 %0, %0\n\tv_rndne_f64 %1, %1\n\tv_rndne_f64 %2, %2\n\tv_rndne_f64 %3, %3\n\tv_rndne_f64 %4, %4\n\tv_rndne_f64 %5, %5\n\tv_rndne_f64 %6, %6\n\tv_rndne_f64 %7, %7" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(bb), "v"(b));
            if constexpr (KIND == 55) asm volatile("v_add_f64 %0, %0, %8\n\tv_add_f64 %1, %1, %8\n\tv_add_f64 %2, %2, %8\n\tv_add_f64 %3, %3, %8\n\tv_add_f64 %4, %4, %8\n\tv_add_f64 %5, %5, %8\n\tv_add_f64 %6, %6, %8\n\tv_add_f64 %7, %7, %8" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(bb), "v"(b));
            if constexpr (KIND == 56) asm volatile("v_cvt_f64_f32 %0, %8\n\tv_cvt_f64_f32 %1, %9\n\tv_cvt_f64_f32 %2, %10\n\tv_cvt_f64_f32 %3, %11\n\tv_cvt_f64_f32 %4, %12\n\tv_cvt_f64_f32 %5, %13\n\tv_cvt_f64_f32 %6, %14\n\tv_cvt_f64_f32 %7, %15" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(bb), "v"(b));
            if constexpr (KIND == 57) asm volatile("v_cvt_f32_f64 %8, %0\n\tv_cvt_f32_f64 %9, %1\n\tv_cvt_f32_f64 %10, %2\n\tv_cvt_f32_f64 %11, %3\n\tv_cvt_f32_f64 %12, %4\n\tv_cvt_f32_f64 %13, %5\n\tv_cvt_f32_f64 %14, %6\n\tv_cvt_f32_f64 %15, %7" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(bb), "v"(b));
            if constexpr (KIND == 58) asm volatile("v_cvt_i32_f64 %8, %0\n\tv_cvt_i32_f64 %9, %1\n\tv_cvt_i32_f64 %10, %2\n\tv_cvt_i32_f64 %11, %3\n\tv_cvt_i32_f64 %12, %4\n\tv_cvt_i32_f64 %13, %5\n\tv_cvt_i32_f64 %14, %6\n\tv_cvt_i32_f64 %15, %7" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(bb), "v"(b));
            if constexpr (KIND == 59) asm volatile("v_cmp_gt_f64 vcc, %0, %8\n\tv_cmp_gt_f64 vcc, %1, %8\n\tv_cmp_gt_f64 vcc, %2, %8\n\tv_cmp_gt_f64 vcc, %3, %8\n\tv_cmp_gt_f64 vcc, %4, %8\n\tv_cmp_gt_f64 vcc, %5, %8\n\tv_cmp_gt_f64 vcc, %6, %8\n\tv_cmp_gt_f64 vcc, %7, %8" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(bb) : "vcc");
            if constexpr (KIND == 60) asm volatile("v_mov_b64 %0, %8\n\tv_mov_b64 %1, %8\n\tv_mov_b64 %2, %8\n\tv_mov_b64 %3, %8\n\tv_mov_b64 %4, %8\n\tv_mov_b64 %5, %8\n\tv_mov_b64 %6, %8\n\tv_mov_b64 %7, %8" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(bb), "v"(b));
        }
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) s += x[k] + p[k].x + p[k].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int KIND>
static void run(float *d, const char *name, int per_block, int waves_per_simd, hipEvent_t e0, hipEvent_t e1) {
    const unsigned iters = 1024;
    const unsigned blocks = 256 * 4 * waves_per_simd;
    k_ops<KIND><<<blocks, 64>>>(d, iters, 1.0001f, 0.5f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) k_ops<KIND><<<blocks, 64>>>(d, iters, 1.0001f, 0.5f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / 5;
    const double instrs = (double)iters * 8 * per_block * waves_per_simd;      // per SIMD
    printf("%-28s waves/simd=%d  %8.1f us  %.2f cycles per instruction per SIMD (at 2.4 GHz)\n", name, waves_per_simd, us, us * 2400.0 / instrs);
}

int main() {
    float *d;
    CK(hipMalloc(&d, 256 * 4 * 16 * 64 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    run<0>(d, "(warm-up)", 8, 8, e0, e1);
    for (int w : {1, 2, 4, 8}) {
        run<0>(d, "v_add_f32 e32", 8, w, e0, e1);
        run<1>(d, "v_mul_f32 e32", 8, w, e0, e1);
        run<2>(d, "v_sub_f32 e32", 8, w, e0, e1);
        run<3>(d, "v_add_f32 e64 (neg mod)", 8, w, e0, e1);
        run<4>(d, "v_add_f32 v, sgpr (e32)", 8, w, e0, e1);
        run<5>(d, "v_add_f32 v, literal", 8, w, e0, e1);
        run<6>(d, "v_min_f32 e32", 8, w, e0, e1);
        run<7>(d, "v_max_f32 e32", 8, w, e0, e1);
        run<8>(d, "v_fma_f32", 8, w, e0, e1);
        run<9>(d, "v_med3_f32", 8, w, e0, e1);
        run<10>(d, "v_pk_add_f32 v,v", 8, w, e0, e1);
        run<11>(d, "v_pk_mul_f32 v,v", 8, w, e0, e1);
        run<12>(d, "v_pk_fma_f32 v,v,v", 8, w, e0, e1);
        run<13>(d, "v_cndmask_b32 e32 vcc", 8, w, e0, e1);
        run<14>(d, "v_cndmask_b32 e64 sgpr", 8, w, e0, e1);
        run<15>(d, "v_cmp_lt_f32 e32 ->vcc", 8, w, e0, e1);
        run<16>(d, "v_cmp_lt_u32 e32 ->vcc", 8, w, e0, e1);
        run<17>(d, "v_cmp_lt_u32 e64 ->sgpr", 8, w, e0, e1);
        run<18>(d, "v_cmp + v_cndmask via vcc", 16, w, e0, e1);
        run<19>(d, "v_add_u32 e32", 8, w, e0, e1);
        run<20>(d, "v_sub_co_u32 e32 ->vcc", 8, w, e0, e1);
        run<21>(d, "v_and_b32 e32", 8, w, e0, e1);
        run<22>(d, "v_xor_b32 e32", 8, w, e0, e1);
        run<23>(d, "v_bfi_b32", 8, w, e0, e1);
        run<24>(d, "v_lshrrev_b32 e32", 8, w, e0, e1);
        run<25>(d, "v_ashrrev_i32 e32", 8, w, e0, e1);
        run<26>(d, "v_mov_b32", 8, w, e0, e1);
        run<27>(d, "v_cvt_f32_u32", 8, w, e0, e1);
        run<28>(d, "v_lshl_add_u32", 8, w, e0, e1);
        run<29>(d, "v_and_or_b32", 8, w, e0, e1);
        run<30>(d, "s_nop 0 (SALU filler)", 8, w, e0, e1);
        run<31>(d, "v_lshlrev_b64", 8, w, e0, e1);
        run<32>(d, "v_lshrrev_b64", 8, w, e0, e1);
        run<33>(d, "v_lshl_add_u64", 8, w, e0, e1);
        run<34>(d, "v_cmp_gt_u64 ->vcc", 8, w, e0, e1);
        run<35>(d, "v_alignbit_b32", 8, w, e0, e1);
        run<36>(d, "v_ffbh_u32", 8, w, e0, e1);
        run<37>(d, "v_min3_u32", 8, w, e0, e1);
        run<38>(d, "v_add_co + v_addc_co pair", 8, w, e0, e1);
        run<39>(d, "v_add_u32 e64 clamp", 8, w, e0, e1);
        run<40>(d, "v_mul_lo_u32", 8, w, e0, e1);
        run<41>(d, "v_mul_f64", 8, w, e0, e1);
        run<42>(d, "dependent v_add_f32 chain", 8, w, e0, e1);
        run<43>(d, "dependent v_xor_b32 chain", 8, w, e0, e1);
        run<44>(d, "dependent v_fma_f32 chain", 8, w, e0, e1);
        run<45>(d, "dependent v_lshlrev_b64 chain", 8, w, e0, e1);
        run<46>(d, "dependent v_alignbit chain", 8, w, e0, e1);
        run<47>(d, "2 interleaved dep add chains", 8, w, e0, e1);
        run<48>(d, "dependent mul,add alternating", 8, w, e0, e1);
        run<49>(d, "dependent mul,sub,xor,shift", 8, w, e0, e1);
        run<50>(d, "dependent cmp->cndmask chain", 8, w, e0, e1);
        run<51>(d, "dependent f64 mul,add", 8, w, e0, e1);
        run<52>(d, "dependent mul,add w/ literals", 8, w, e0, e1);
        run<53>(d, "v_fma_f64", 8, w, e0, e1);
        run<54>(d, "v_rndne_f64", 8, w, e0, e1);
        run<55>(d, "v_add_f64", 8, w, e0, e1);
        run<56>(d, "v_cvt_f64_f32", 8, w, e0, e1);
        run<57>(d, "v_cvt_f32_f64", 8, w, e0, e1);
        run<58>(d, "v_cvt_i32_f64", 8, w, e0, e1);
        run<59>(d, "v_cmp_gt_f64 ->vcc", 8, w, e0, e1);
        run<60>(d, "v_mov_b64", 8, w, e0, e1);
    }
    return 0;
}
