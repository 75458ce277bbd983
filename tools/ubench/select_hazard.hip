// GPU box: hipcc -O3 --offload-arch=gfx950 -o tools/ubench/select_hazard tools/ubench/select_hazard.hip && tools/ubench/select_hazard
// Two questions about v_cndmask_b32_e64 with a scalar mask operand, asked after the random-script fuzz found lanes 32-63
// selecting wrongly in one generated kernel:
//  (1) mask operand = the EXEC register itself (what the compiler made of an inline asm's `"s"(ballot(true))`): every
//      active lane should take the second value.  On gfx950 the upper half-wave does not.
//  (2) compare -> select pairs with 0, 1 and 2 wait states (s_nop) between the compare that writes the SGPR pair and the
//      select that reads it, with and without a 64-bit multiply before the compare, one and four waves per workgroup:
//      wrong selects by half of the wave.  None seen.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int NOPS, int PAD>
__global__ void k_sel(const float *__restrict__ x, uint32_t n, uint32_t iters, unsigned long long *bad_lo, unsigned long long *bad_hi) {
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long wrong = 0;
    float acc = 0.0f;
    for (uint32_t it = 0; it < iters; it++) {
        const float v = x[(it * 64 + lane + blockIdx.x * 977) % n];
        const float thr = x[(it * 131 + 7 + blockIdx.x) % n];
        float a = v + 1.0f, b = v - 1.0f, r;
        // a previous, different mask in the same SGPR pair: !(v > thr) -- a stale read selects the other operand
        if (PAD == 0) {
            if (NOPS == 0) asm volatile("v_cmp_ngt_f32_e64 s[20:21], %1, %2\n\ts_nop 4\n\tv_cmp_gt_f32_e64 s[20:21], %1, %2\n\tv_cndmask_b32_e64 %0, %3, %4, s[20:21]" : "=v"(r) : "v"(v), "v"(thr), "v"(b), "v"(a) : "s20", "s21");
            if (NOPS == 1) asm volatile("v_cmp_ngt_f32_e64 s[20:21], %1, %2\n\ts_nop 4\n\tv_cmp_gt_f32_e64 s[20:21], %1, %2\n\ts_nop 0\n\tv_cndmask_b32_e64 %0, %3, %4, s[20:21]" : "=v"(r) : "v"(v), "v"(thr), "v"(b), "v"(a) : "s20", "s21");
            if (NOPS == 2) asm volatile("v_cmp_ngt_f32_e64 s[20:21], %1, %2\n\ts_nop 4\n\tv_cmp_gt_f32_e64 s[20:21], %1, %2\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %3, %4, s[20:21]" : "=v"(r) : "v"(v), "v"(thr), "v"(b), "v"(a) : "s20", "s21");
        } else {
            // independent VALU work right before the compare (a 64-bit op, as in the kernel that failed)
            double d = (double)v;
            if (NOPS == 0) asm volatile("v_cmp_ngt_f32_e64 s[20:21], %2, %3\n\ts_nop 4\n\tv_mul_f64 %1, %1, %1\n\tv_cmp_gt_f32_e64 s[20:21], %2, %3\n\tv_cndmask_b32_e64 %0, %4, %5, s[20:21]" : "=v"(r), "+v"(d) : "v"(v), "v"(thr), "v"(b), "v"(a) : "s20", "s21");
            if (NOPS == 1) asm volatile("v_cmp_ngt_f32_e64 s[20:21], %2, %3\n\ts_nop 4\n\tv_mul_f64 %1, %1, %1\n\tv_cmp_gt_f32_e64 s[20:21], %2, %3\n\ts_nop 0\n\tv_cndmask_b32_e64 %0, %4, %5, s[20:21]" : "=v"(r), "+v"(d) : "v"(v), "v"(thr), "v"(b), "v"(a) : "s20", "s21");
            if (NOPS == 2) asm volatile("v_cmp_ngt_f32_e64 s[20:21], %2, %3\n\ts_nop 4\n\tv_mul_f64 %1, %1, %1\n\tv_cmp_gt_f32_e64 s[20:21], %2, %3\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %4, %5, s[20:21]" : "=v"(r), "+v"(d) : "v"(v), "v"(thr), "v"(b), "v"(a) : "s20", "s21");
            acc += (float)d * 1e-30f;
        }
        const float want = v > thr ? a : b;
        wrong += (__float_as_uint(r) != __float_as_uint(want));
        acc += r;
    }
    if (acc == 12345.678f) wrong += 1u << 30;                       // keep acc alive
    atomicAdd(lane < 32 ? bad_lo : bad_hi, wrong);
}

// v_cndmask_b32_e64 with EXEC itself as the mask operand: what the compiler made of `"s"(ballot(true))` in the kernel that
// failed.  Every active lane should take `a`.
__global__ void k_sel_exec(const float *__restrict__ x, uint32_t n, unsigned long long *bad_lo, unsigned long long *bad_hi) {
    const uint32_t lane = threadIdx.x & 63u;
    const float v = x[(lane + blockIdx.x * 977) % n];
    float a = v + 1.0f, b = v - 1.0f, r, r2;
    asm volatile("v_cndmask_b32_e64 %0, %1, %2, exec" : "=v"(r) : "v"(b), "v"(a));
    asm volatile("s_mov_b64 s[20:21], exec\n\ts_nop 4\n\tv_cndmask_b32_e64 %0, %1, %2, s[20:21]" : "=v"(r2) : "v"(b), "v"(a) : "s20", "s21");
    atomicAdd(lane < 32 ? bad_lo : bad_hi, (unsigned long long)(__float_as_uint(r) != __float_as_uint(a)) + ((unsigned long long)(__float_as_uint(r2) != __float_as_uint(a)) << 32));
}

template <int NOPS, int PAD> static void run(const float *dx, uint32_t n, unsigned long long *dbad, uint32_t waves_per_block) {
    hipMemset(dbad, 0, 16);
    const uint32_t iters = 20000, blocks = 1024;
    hipLaunchKernelGGL((k_sel<NOPS, PAD>), dim3(blocks), dim3(64 * waves_per_block), 0, 0, dx, n, iters, dbad, dbad + 1);
    unsigned long long bad[2];
    hipMemcpy(bad, dbad, 16, hipMemcpyDeviceToHost);
    printf("wait states %d, %s, %u wave(s) per workgroup: wrong selects lanes 0-31 %llu, lanes 32-63 %llu of %llu each\n", NOPS,
           PAD ? "v_mul_f64 before the compare" : "compare first", waves_per_block, bad[0], bad[1], (unsigned long long)blocks * waves_per_block * 32 * iters);
}

int main() {
    const uint32_t n = 1 << 16;
    std::vector<float> x(n);
    uint32_t s = 12345;
    for (auto &v : x) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) / 16777216.0f * 4.0f - 2.0f; }
    float *dx; unsigned long long *dbad;
    hipMalloc(&dx, n * 4); hipMalloc(&dbad, 16);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    {
        hipMemset(dbad, 0, 16);
        hipLaunchKernelGGL(k_sel_exec, dim3(1024), dim3(64), 0, 0, dx, n, dbad, dbad + 1);
        unsigned long long bad[2];
        hipMemcpy(bad, dbad, 16, hipMemcpyDeviceToHost);
        printf("mask operand = exec itself: wrong selects lanes 0-31 %llu, lanes 32-63 %llu of %d each;  mask = a copy of exec in s[20:21]: %llu, %llu\n",
               bad[0] & 0xffffffffull, bad[1] & 0xffffffffull, 1024 * 32, bad[0] >> 32, bad[1] >> 32);
    }
    for (uint32_t w : {1u, 4u}) {
        run<0, 0>(dx, n, dbad, w); run<1, 0>(dx, n, dbad, w); run<2, 0>(dx, n, dbad, w);
        run<0, 1>(dx, n, dbad, w); run<1, 1>(dx, n, dbad, w); run<2, 1>(dx, n, dbad, w);
    }
    return 0;
}
