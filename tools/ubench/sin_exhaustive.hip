// sin_exhaustive.hip -- is a cheaper evaluation of musl's sinf / cosf BIT-IDENTICAL to the library's for EVERY f32?
//
// The parity contract is the bits of (float)(f64 polynomial of the f64-reduced argument); the operation sequence is only
// a means.  A fused multiply-add changes an f64 intermediate by at most one rounding, and the f32 result only when that
// lands on a rounding boundary of the final conversion -- expected 2^-11 .. 1 times over all 2^32 arguments depending on
// which operation is fused.  This program decides by running all 2^32 bit patterns through the library's zsinf / zcosf
// (zmath.hip.h, the forms the oracle tests pin) and through each candidate, and counting differing results.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -I../../zang_amd/csrc sin_exhaustive.hip -o /tmp/sin_ex && /tmp/sin_ex
//
// Candidate bits (a candidate = a set of them):
//   1  REDM1   ym's first step as fma(-fn, pio2_1, xd)        (fn * pio2_1 is exact below 2^28: provably the same bits)
//   2  REDM2   ym's second step as fma(-fn, pio2_1t, .)
//   4  TAILH   S3 + z*S4 and C2 + z*C3 as fma
//   8  TAILA   the final add of each kernel as fma(Cq, T, head)
//  16  COSW    cosdf: (1 + z*C0) + w*C1 as fma(w, C1, 1 + z*C0)
//  32  SINP    sindf: y + s*P as fma(s, P, y)
//  64  SINH    sindf: S1 + z*S2 as fma
// 128  COSH    cosdf: 1 + z*C0 as fma(z, C0, 1)
// 256  ONELEAF y = ym for every |x| (drop the small leaf x - fn*pio2)
// 512  YSFMA   the small leaf as fma(-fn, pio2, xd) (3 * pio2 no longer rounded on its own)
// 1024 NOTINY  no early return of x (sin) / 1 (cos) below 2^-12: the polynomial path for every finite argument
// 2048 NOOOR   the rare path only for |x| >= 2^28 pi/2: no |ym| > pi/4 test ("matters with directed rounding", musl)
// 4096 NOINF   (with NOOOR) inf / nan take the rare path's x - x as y instead of a final select
// 8192 MAGIC   fn = fma(xd, invpio2, 1.5*2^52) - 1.5*2^52, n = the low mantissa bits (no rndne, no cvt_i32)
// ONELEAF differs for exactly two arguments, +-0x40406406 (sinf one ulp high, cosf unaffected): the library takes it and
// corrects that magnitude by name (zmath.hip.h kZSinOneLeafOdd); the THE LIBRARY line checks the result.
// The REFERENCE is candidate 0: musl's operation order with every rounding (oracle/zmath_ref.h's order); the library's
// own zsinf / zcosf (whatever zmath.hip.h holds today) is checked against it too.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "zmath.hip.h"

enum { REDM1 = 1, REDM2 = 2, TAILH = 4, TAILA = 8, COSW = 16, SINP = 32, SINH = 64, COSH = 128, ONELEAF = 256, YSFMA = 512, NOTINY = 1024, NOOOR = 2048, NOINF = 4096, MAGIC = 8192, LIBRARY = 1 << 20 };

template <int FL>
__device__ __forceinline__ int cand_reduce(float x, uint32_t ix, double &y) {
    const double invpio2 = 6.36619772367581382433e-01, pio2 = 1.57079632679489661923,
                 pio2_1 = 1.57079631090164184570e+00, pio2_1t = 1.58932547735281966916e-08, pio4 = 0x1.921fb6p-1;
    const double toint = 1.5 / 2.220446049250313e-16;
    const double xd = (double)x;
    double fn;
    int n;
    if (FL & MAGIC) {
        const double fnm = __builtin_fma(xd, invpio2, toint);          // musl's own x*invpio2 + toint - toint, product unrounded
        fn = fnm - toint;
        n = (int)(uint32_t)__double_as_longlong(fnm);                   // low mantissa bits = the integer (two's complement)
    } else {
        fn = __builtin_rint(xd * invpio2);
        n = (int)fn;
    }
    const double m1 = (FL & REDM1) ? __builtin_fma(-fn, pio2_1, xd) : xd - fn * pio2_1;
    const double ym = (FL & REDM2) ? __builtin_fma(-fn, pio2_1t, m1) : m1 - fn * pio2_1t;
    const bool small = ix <= 0x40e231d5;
    if (FL & ONELEAF) y = ym;
    else { const double ys = (FL & YSFMA) ? __builtin_fma(-fn, pio2, xd) : xd - fn * pio2; y = small ? ys : ym; }
    bool rare;
    if (FL & NOOOR) rare = ix > 0x4dc90fdau && ((FL & NOINF) || ix < 0x7f800000u);
    else rare = !small && (__builtin_fabs(ym) > pio4 || (ix - 0x4dc90fdbu) < (0x7f800000u - 0x4dc90fdbu));
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(rare) != 0, 0)) {
        if (rare) n = zrem_pio2f(x, &y);
    }
    return n;
}

template <int FL>
__device__ __forceinline__ void cand_kernels(double y, float &sv, float &cv) {
    const double S1 = -0x15555554cbac77.0p-55, S2 = 0x111110896efbb2.0p-59, S3 = -0x1a00f9e2cae774.0p-65, S4 = 0x16cd878c3b46a7.0p-71;
    const double C0 = -0x1ffffffd0c5e81.0p-54, C1 = 0x155553e1053a42.0p-57, C2 = -0x16c087e80f1e27.0p-62, C3 = 0x199342e0ee5069.0p-68;
    const double z = y * y, w = z * z, s = z * y;
    const double sp = (FL & SINH) ? __builtin_fma(z, S2, S1) : S1 + z * S2;
    const double sh = (FL & SINP) ? __builtin_fma(s, sp, y) : y + s * sp;
    const double st = (FL & TAILH) ? __builtin_fma(z, S4, S3) : S3 + z * S4;
    const double sq = s * w;
    sv = (float)((FL & TAILA) ? __builtin_fma(sq, st, sh) : sh + sq * st);
    const double c1 = (FL & COSH) ? __builtin_fma(z, C0, 1.0) : 1.0 + z * C0;
    const double ch = (FL & COSW) ? __builtin_fma(w, C1, c1) : c1 + w * C1;
    const double ct = (FL & TAILH) ? __builtin_fma(z, C3, C2) : C2 + z * C3;
    const double cq = w * z;
    cv = (float)((FL & TAILA) ? __builtin_fma(cq, ct, ch) : ch + cq * ct);
}

template <int FL>
__device__ __forceinline__ float cand_sinf(float x) {
    const uint32_t ix = zf2u(x) & 0x7fffffff;
    double y;
    const int n = cand_reduce<FL>(x, ix, y);
    float sv, cv;
    cand_kernels<FL>(y, sv, cv);
    float r = zu2f(zf2u((n & 1) ? cv : sv) ^ ((uint32_t)(n & 2) << 30));
    if (!(FL & NOTINY) && ix < 0x39800000) r = x;
    if (!(FL & NOINF) && ix >= 0x7f800000) r = x - x;
    return r;
}
template <int FL>
__device__ __forceinline__ float cand_cosf(float x) {
    const uint32_t ix = zf2u(x) & 0x7fffffff;
    double y;
    const int n = cand_reduce<FL>(x, ix, y);
    float sv, cv;
    cand_kernels<FL>(y, sv, cv);
    float r = zu2f(zf2u((n & 1) ? sv : cv) ^ ((uint32_t)((n + 1) & 2) << 30));
    if (!(FL & NOTINY) && ix < 0x39800000) r = 1.0f;
    if (!(FL & NOINF) && ix >= 0x7f800000) r = x - x;
    return r;
}

struct Result { unsigned long long bad_sin, bad_cos; uint32_t first_sin, first_cos; uint32_t list[16], want[16], got[16]; };

// one launch covers `count` consecutive bit patterns from `first`
template <int FL>
__global__ void __launch_bounds__(256) k_check(uint32_t first, Result *res) {
    const uint32_t u = first + blockIdx.x * 256u + threadIdx.x;
    const float x = zu2f(u);
    const uint32_t a = zf2u(cand_sinf<0>(x)), b = zf2u(FL == LIBRARY ? zsinf(x) : cand_sinf<FL & 0xfffff>(x));
    const uint32_t c = zf2u(cand_cosf<0>(x)), d = zf2u(FL == LIBRARY ? zcosf(x) : cand_cosf<FL & 0xfffff>(x));
    // NaN results: reference and candidate both return x - x; payloads are compared too
    if (a != b) { const unsigned long long k = atomicAdd(&res->bad_sin, 1ull); atomicMin(&res->first_sin, u); if (k < 16) { res->list[k] = u; res->want[k] = a; res->got[k] = b; } }
    if (c != d) { atomicAdd(&res->bad_cos, 1ull); atomicMin(&res->first_cos, u); }
}

template <int FL>
static Result run(const char *name, bool quick) {
    Result *dev, host{0, 0, 0xffffffffu, 0xffffffffu, {0}, {0}, {0}};
    hipMalloc(&dev, sizeof(Result));
    hipMemcpy(dev, &host, sizeof host, hipMemcpyHostToDevice);
    const uint32_t chunk = 1u << 28;                         // 16 launches of 2^28 patterns
    for (uint32_t k = 0; k < 16; k++) {
        if (quick && (k & 3)) continue;
        hipLaunchKernelGGL(k_check<FL>, dim3(chunk / 256), dim3(256), 0, 0, k * chunk, dev);
    }
    hipDeviceSynchronize();
    hipMemcpy(&host, dev, sizeof host, hipMemcpyDeviceToHost);
    hipFree(dev);
    printf("%-44s flags %7d: sin %llu differing", name, FL, host.bad_sin);
    if (host.bad_sin) { printf(" (first x = 0x%08x;", host.first_sin); for (unsigned long long k = 0; k < host.bad_sin && k < 16; k++) printf(" x=0x%08x want 0x%08x got 0x%08x", host.list[k], host.want[k], host.got[k]); printf(")"); }
    printf(", cos %llu differing", host.bad_cos);
    if (host.bad_cos) printf(" (first x = 0x%08x)", host.first_cos);
    printf("\n");
    fflush(stdout);
    return host;
}

int main(int argc, char **argv) {
    const bool quick = argc > 1 && atoi(argv[1]) == 1;      // 1 = a quarter of the patterns (smoke)
    printf("all %s f32 bit patterns, musl's operation order (candidate 0) vs candidate\n", quick ? "2^30 (quick)" : "2^32");
    run<0>("identity (the checker itself)", quick);
    run<LIBRARY>("THE LIBRARY: zmath.hip.h zsinf / zcosf", quick);
    run<REDM1>("REDM1 fma(-fn, pio2_1, x)", quick);
    run<REDM2>("REDM2 fma(-fn, pio2_1t, .)", quick);
    run<TAILH>("TAILH S3+z*S4, C2+z*C3 fused", quick);
    run<TAILA>("TAILA final adds fused", quick);
    run<COSW>("COSW fma(w, C1, 1+z*C0)", quick);
    run<SINP>("SINP fma(s, S1+z*S2, y)", quick);
    run<SINH>("SINH fma(z, S2, S1)", quick);
    run<COSH>("COSH fma(z, C0, 1)", quick);
    run<ONELEAF>("ONELEAF y = ym everywhere", quick);
    run<YSFMA>("YSFMA fma(-fn, pio2, x)", quick);
    run<NOTINY>("NOTINY no |x| < 2^-12 early return", quick);
    run<NOTINY | 767>("NOTINY + all nine fusions", quick);
    run<NOOOR>("NOOOR", quick);
    run<NOOOR | NOINF>("NOOOR+NOINF", quick);
    run<MAGIC>("MAGIC", quick);
    run<ONELEAF | REDM1>("ONELEAF+REDM1", quick);
    run<ONELEAF | REDM2>("ONELEAF+REDM2", quick);
    run<ONELEAF | REDM1 | REDM2>("ONELEAF+REDM1+REDM2", quick);
    run<ONELEAF | MAGIC>("ONELEAF+MAGIC", quick);
    run<ONELEAF | MAGIC | REDM1 | REDM2>("ONELEAF+MAGIC+REDM1+REDM2", quick);
    run<767 | NOOOR | NOINF | MAGIC>("nine fusions + NOOOR+NOINF+MAGIC", quick);
    run<767 | NOOOR | NOINF | MAGIC | ONELEAF>("nine fusions + NOOOR+NOINF+MAGIC+ONELEAF", quick);
    run<REDM1 | REDM2 | TAILH | TAILA>("REDM1+REDM2+TAILH+TAILA", quick);
    run<REDM1 | REDM2 | TAILH | TAILA | COSW>("... +COSW", quick);
    run<REDM1 | REDM2 | TAILH | TAILA | COSW | SINP>("... +COSW+SINP", quick);
    run<REDM1 | REDM2 | TAILH | TAILA | COSW | SINP | SINH>("... +COSW+SINP+SINH", quick);
    run<REDM1 | REDM2 | TAILH | TAILA | COSW | SINP | SINH | COSH>("... +COSW+SINP+SINH+COSH", quick);
    run<REDM1 | REDM2 | TAILH | TAILA | COSW | SINP | SINH | COSH | YSFMA>("... +COSW+SINP+SINH+COSH+YSFMA", quick);
    return 0;
}
