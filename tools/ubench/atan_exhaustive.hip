// atan_exhaustive.hip -- is a cheaper evaluation of musl's atanf BIT-IDENTICAL to the reference order for EVERY f32?  (VERDICT r3 item 7;
// the sweep that decided zsinf / zcosf in round 3, tools/ubench/sin_exhaustive.hip, for Distortion's overdrive: Distortion.zig:41-52
// calls std.math.atan per sample.)
//
// Unlike sinf, musl's atanf works in f32 throughout: there is no f64 slack for a fused multiply-add to hide in, so most fusions are
// expected to differ somewhere.  What can be exact: forms whose products are exact (2x, 0.5x, 1x), one divide for all ranges, a
// cheaper divide that is still correctly rounded on the ranges atanf feeds it, selects replaced by a per-range coefficient row.
// This program runs all 2^32 bit patterns through musl's own branchy order (candidate 0) and through each candidate and counts
// differing results (NaN payloads included).
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -I../../zang_amd/csrc atan_exhaustive.hip -o atan_exhaustive && ./atan_exhaustive
//
// Candidate bits:
//    1 FOLD     the round-2 library form: every range's numerator / denominator by selects, ONE divide, selects for the tails
//    2 UNIFIED  num = |x| - c, den = 1 + c |x| for c = 0.5 / 1 / 1.5 (the first range scaled by an exact 1/2: same quotient bits), -1 / |x| selected
//    4 ROWS     no range selects at all: num = fma(a, |x|, b), den = d + c |x| with one coefficient row (a, b, c, d, hi, lo) per
//               range picked by an index -- including the direct range |x| < 7/16 as the row (1, 0, 0, 1, 0, 0): x / 1, 0 - ((t - 0) - x)
//    8 FASTDIV  the divide as rcp + one Newton step on the reciprocal + one residual correction of the quotient (no v_div_scale /
//               v_div_fmas / v_div_fixup: the operands are far from the exponent range's ends here)
//   16 RCP1     ... without the Newton step on the reciprocal (one residual correction only)
//   32 POLY_A   aT2 + w aT4 fused      64 POLY_B  aT0 + w (.) fused     128 POLY_C  aT1 + w aT3 fused
//  256 NOTINY   no early return of x below 2^-12
//  512 INDEXSUM the row index as a sum of sign bits ((k - 1 - ix) >> 31) instead of compares (same integers: checked through ROWS)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "zmath.hip.h"

enum { FOLD = 1, UNIFIED = 2, ROWS = 4, FASTDIV = 8, RCP1 = 16, POLY_A = 32, POLY_B = 64, POLY_C = 128, NOTINY = 256, INDEXSUM = 512, LIBRARY = 1 << 20 };

__device__ __forceinline__ float musl_atanf(float x) {                // musl src/math/atanf.c, operation for operation
    const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT[5] = {3.3333328366e-01f, -1.9999158382e-01f, 1.4253635705e-01f, -1.0648017377e-01f, 6.1687607318e-02f};
    uint32_t ix = zf2u(x);
    const uint32_t sign = ix >> 31;
    ix &= 0x7fffffff;
    int id;
    if (ix >= 0x4c800000) {
        if (x != x) return x;
        const float z = atanhi[3] + 0x1p-120f;
        return sign ? -z : z;
    }
    if (ix < 0x3ee00000) {
        if (ix < 0x39800000) return x;
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else { id = 1; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else { id = 3; x = -1.0f / x; }
        }
    }
    float z = x * x;
    const float w = z * z;
    const float s1 = z * (aT[0] + w * (aT[2] + w * aT[4]));
    const float s2 = w * (aT[1] + w * aT[3]);
    if (id < 0) return x - x * (s1 + s2);
    z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return sign ? -z : z;
}

template <int FL>
__device__ __forceinline__ float cand_div(float num, float den) {
    if (FL & (FASTDIV | RCP1)) {
        float r = __builtin_amdgcn_rcpf(den);
        if (!(FL & RCP1)) r = __builtin_fmaf(__builtin_fmaf(-den, r, 1.0f), r, r);
        const float q = num * r;
        return __builtin_fmaf(__builtin_fmaf(-den, q, num), r, q);
    }
    return num / den;
}

template <int FL>
__device__ __forceinline__ float cand_atanf(float x) {
    const float aT0 = 3.3333328366e-01f, aT1 = -1.9999158382e-01f, aT2 = 1.4253635705e-01f, aT3 = -1.0648017377e-01f, aT4 = 6.1687607318e-02f;
    const uint32_t ux = zf2u(x), ix = ux & 0x7fffffff;
    const bool sign = (ux >> 31) != 0;
    const float ax = fabsf(x);
    float xr, hi, lo;
    bool direct = ix < 0x3ee00000;
    if (FL & ROWS) {
        // rows 0..4: |x| < 7/16, < 11/16, < 19/16, < 39/16, the rest
        uint32_t id;
        if (FL & INDEXSUM) id = ((0x3edfffffu - ix) >> 31) + ((0x3f2fffffu - ix) >> 31) + ((0x3f97ffffu - ix) >> 31) + ((0x401bffffu - ix) >> 31);
        else id = (ix >= 0x3ee00000) + (ix >= 0x3f300000) + (ix >= 0x3f980000) + (ix >= 0x401c0000);
        const float A[5] = {1.0f, 1.0f, 1.0f, 1.0f, 0.0f}, B[5] = {0.0f, -0.5f, -1.0f, -1.5f, -1.0f};
        const float Cc[5] = {0.0f, 0.5f, 1.0f, 1.5f, 1.0f}, D[5] = {1.0f, 1.0f, 1.0f, 1.0f, 0.0f};
        const float HI[5] = {0.0f, 4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
        const float LO[5] = {0.0f, 5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
        const float num = __builtin_fmaf(A[id], ax, B[id]);
        const float den = D[id] + Cc[id] * ax;
        xr = cand_div<FL>(num, den);
        hi = HI[id]; lo = LO[id];
        direct = false;                                                // the direct range is row 0
    } else {
        const bool r0 = ix < 0x3f300000, r1 = ix < 0x3f980000, r2 = ix < 0x401c0000;
        float num, den;
        if (FL & UNIFIED) {
            const float c = r0 ? 0.5f : (r1 ? 1.0f : 1.5f);
            num = r2 ? ax - c : -1.0f;
            den = r2 ? 1.0f + c * ax : ax;
        } else {
            num = r1 ? (r0 ? 2.0f * ax - 1.0f : ax - 1.0f) : (r2 ? ax - 1.5f : -1.0f);
            den = r1 ? (r0 ? 2.0f + ax : ax + 1.0f) : (r2 ? 1.0f + 1.5f * ax : ax);
        }
        hi = r1 ? (r0 ? 4.6364760399e-01f : 7.8539812565e-01f) : (r2 ? 9.8279368877e-01f : 1.5707962513e+00f);
        lo = r1 ? (r0 ? 5.0121582440e-09f : 3.7748947079e-08f) : (r2 ? 3.4473217170e-08f : 7.5497894159e-08f);
        xr = direct ? x : cand_div<FL>(num, den);
    }
    const float z = xr * xr;
    const float w = z * z;
    const float pa = (FL & POLY_A) ? __builtin_fmaf(w, aT4, aT2) : aT2 + w * aT4;
    const float pb = (FL & POLY_B) ? __builtin_fmaf(w, pa, aT0) : aT0 + w * pa;
    const float s1 = z * pb;
    const float pc = (FL & POLY_C) ? __builtin_fmaf(w, aT3, aT1) : aT1 + w * aT3;
    const float s2 = w * pc;
    const float t = xr * (s1 + s2);
    const float zz = hi - ((t - lo) - xr);
    float r = direct ? xr - t : (sign ? -zz : zz);
    if (!(FL & NOTINY) && ix < 0x39800000) r = x;
    if (ix >= 0x4c800000) {
        const float big = 1.5707962513e+00f + 0x1p-120f;
        r = (x != x) ? x : (sign ? -big : big);
    }
    return r;
}

struct Result { unsigned long long bad; uint32_t first; uint32_t list[8], want[8], got[8]; };

template <int FL>
__global__ void __launch_bounds__(256) k_check(uint32_t first, Result *res) {
    const uint32_t u = first + blockIdx.x * 256u + threadIdx.x;
    const float x = zu2f(u);
    const uint32_t a = zf2u(musl_atanf(x)), b = zf2u(FL == LIBRARY ? zatanf(x) : cand_atanf<FL & 0xfffff>(x));
    if (a != b) { const unsigned long long k = atomicAdd(&res->bad, 1ull); atomicMin(&res->first, u); if (k < 8) { res->list[k] = u; res->want[k] = a; res->got[k] = b; } }
}

// time of 2^28 evaluations (the sum is stored so that nothing is optimised away)
template <int FL>
__global__ void __launch_bounds__(256) k_time(uint32_t first, float *out) {
    const uint32_t u = first + (blockIdx.x * 256u + threadIdx.x) * 16u;
    float s = 0.0f;
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) {
        const float x = zu2f((u + k) & 0x41ffffffu);                   // finite, |x| up to 32: every range
        s += FL == LIBRARY ? zatanf(x) : (FL == -1 ? musl_atanf(x) : cand_atanf<FL & 0xfffff>(x));
    }
    out[blockIdx.x * 256u + threadIdx.x] = s;
}

template <int FL>
static void run(const char *name, bool quick) {
    Result *dev, host{0, 0xffffffffu, {0}, {0}, {0}};
    hipMalloc(&dev, sizeof(Result));
    hipMemcpy(dev, &host, sizeof host, hipMemcpyHostToDevice);
    const uint32_t chunk = 1u << 28;
    for (uint32_t k = 0; k < 16; k++) {
        if (quick && (k & 3)) continue;
        hipLaunchKernelGGL(k_check<FL>, dim3(chunk / 256), dim3(256), 0, 0, k * chunk, dev);
    }
    hipDeviceSynchronize();
    hipMemcpy(&host, dev, sizeof host, hipMemcpyDeviceToHost);
    hipFree(dev);
    float *out;
    hipMalloc(&out, (size_t)(1u << 24) * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_time<FL>, dim3((1u << 24) / 256), dim3(256), 0, 0, 0u, out);
    hipEventRecord(e0, 0);
    for (int rep = 0; rep < 4; rep++) hipLaunchKernelGGL(k_time<FL>, dim3((1u << 24) / 256), dim3(256), 0, 0, 0x3c000000u, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    printf("%-58s flags %7d: %llu differing", name, FL, host.bad);
    if (host.bad) { printf(" (first x = 0x%08x;", host.first); for (unsigned long long k = 0; k < host.bad && k < 3; k++) printf(" x=0x%08x want 0x%08x got 0x%08x", host.list[k], host.want[k], host.got[k]); printf(")"); }
    printf("   | %.3f ms per 2^28 evaluations (%.2f ps each)\n", ms / 4.0, ms * 1e9 / 4.0 / (double)(1u << 28));
    fflush(stdout);
}

int main(int argc, char **argv) {
    const bool quick = argc > 1 && atoi(argv[1]) == 1;
    printf("all %s f32 bit patterns, musl's atanf (branches, its own operation order) vs candidate\n", quick ? "2^30 (quick)" : "2^32");
    run<0>("selects, one divide (no flags)", quick);
    run<LIBRARY>("THE LIBRARY: zmath.hip.h zatanf", quick);
    run<FOLD>("FOLD (the round-2 form)", quick);
    run<UNIFIED>("UNIFIED |x| - c over 1 + c |x|", quick);
    run<ROWS>("ROWS coefficient row per range, direct range folded", quick);
    run<ROWS | INDEXSUM>("ROWS+INDEXSUM", quick);
    run<FASTDIV>("FASTDIV rcp + Newton + residual", quick);
    run<RCP1>("RCP1 rcp + residual", quick);
    run<UNIFIED | FASTDIV>("UNIFIED+FASTDIV", quick);
    run<UNIFIED | RCP1>("UNIFIED+RCP1 (the library's form since round 4)", quick);
    run<ROWS | FASTDIV>("ROWS+FASTDIV", quick);
    run<ROWS | INDEXSUM | FASTDIV>("ROWS+INDEXSUM+FASTDIV", quick);
    run<ROWS | INDEXSUM | RCP1>("ROWS+INDEXSUM+RCP1", quick);
    run<POLY_A>("POLY_A fma(w, aT4, aT2)", quick);
    run<POLY_B>("POLY_B fma(w, ., aT0)", quick);
    run<POLY_C>("POLY_C fma(w, aT3, aT1)", quick);
    run<NOTINY>("NOTINY no |x| < 2^-12 early return", quick);
    run<ROWS | NOTINY>("ROWS+NOTINY", quick);
    run<ROWS | NOTINY | FASTDIV>("ROWS+NOTINY+FASTDIV", quick);
    run<ROWS | NOTINY | RCP1>("ROWS+NOTINY+RCP1", quick);
    run<ROWS | NOTINY | RCP1 | INDEXSUM>("ROWS+NOTINY+RCP1+INDEXSUM", quick);
    return 0;
}
