// zmath.hip.h -- device scalar math for the paint kernels (gfx950).
//
// The reference takes sin/cos/atan/pow/floor/round and its PRNG from the Zig standard
// library (call sites: SineOsc.zig:5, Filter.zig:21-22, Distortion.zig:41-50,
// Noise.zig:22-58), which documents those routines as ports of musl libc (and of Go's
// math.Pow).  To reproduce the reference's bits rather than "a sine", the kernels run the
// same published algorithms: f64 polynomial kernels after a Cody-Waite / Payne-Hanek
// reduction for sin/cos, the f32 table+polynomial atanf, frexp/ldexp square-and-multiply
// pow, xoshiro256++.  An f64 VALU operation issues in the 4.2-cycle class on gfx950 (measured,
// profiles/r01/ubench), a plain f32 add / mul in 2.5: one sinf was ~50 instructions, 35 of them f64 (24 since round 3:
// multiply-add pairs fused where ALL 2^32 arguments give the same bits, zsincos_kernels below).
// What can be changed without changing bits is the control flow: a wave's lanes sit in every range
// of a libm routine at once, so the ranges are folded into straight-line code in which each lane
// performs exactly its range's operations (sinf / cosf, atanf below).
//
// Everything here must be compiled with -ffp-contract=off (no fused multiply-add by the compiler: Zig
// emits none for these loops; the explicit __builtin_fma calls below are the exhaustively verified exceptions)
// and without fast-math.
#pragma once
#include <hip/hip_runtime.h>
#if !defined(__HIPCC_RTC__)
#include <stdint.h>
#endif
#include "rtc_types.hip.h"

#define ZD __device__ __forceinline__

ZD uint32_t zf2u(float f) { return __float_as_uint(f); }
ZD float zu2f(uint32_t u) { return __uint_as_float(u); }

// @intFromFloat with NaN / out-of-range DEFINED as v_cvt_u32_f32 / v_cvt_i32_f32 behave (NaN -> 0, saturating).  On the
// device it IS that one instruction: written out as C++ the three guards became three exec-mask branches per conversion
// (the controlled-frequency PulseOsc and the Sampler convert once or twice per sample).
ZD uint32_t zf32_to_u32(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
#else
    if (!(v == v)) return 0u;
    if (v <= 0.0f) return 0u;
    if (v >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)v;
#endif
}
ZD int32_t zf32_to_i32(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    int32_t r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
#else
    if (!(v == v)) return 0;
    if (v <= -2147483648.0f) return INT32_MIN;
    if (v >= 2147483648.0f) return INT32_MAX;
    return (int32_t)v;
#endif
}

// std.math.clamp == @max(lo, @min(v, hi)) as explicit selects (first operand wins ties)
ZD float zminf(float a, float b) { return (a <= b || b != b) ? a : b; }
ZD float zmaxf(float a, float b) { return (a >= b || b != b) ? a : b; }
ZD float zclampf(float v, float lo, float hi) { return zmaxf(lo, zminf(v, hi)); }

// ---- sin / cos ----------------------------------------------------------------------
// 24-bit chunks of 2/pi and pi/2 (tools/gen_pio2_tables.py).  Only reached for
// |x| >= 2^28*pi/2; f32 exponents need the first dozen entries at most, the rest serve the
// recomputation loop.
__device__ static const int32_t ZM_IPIO2[66] = {
  0xA2F983, 0x6E4E44, 0x1529FC, 0x2757D1, 0xF534DD, 0xC0DB62,
  0x95993C, 0x439041, 0xFE5163, 0xABDEBB, 0xC561B7, 0x246E3A,
  0x424DD2, 0xE00649, 0x2EEA09, 0xD1921C, 0xFE1DEB, 0x1CB129,
  0xA73EE8, 0x8235F5, 0x2EBB44, 0x84E99C, 0x7026B4, 0x5F7E41,
  0x3991D6, 0x398353, 0x39F49C, 0x845F8B, 0xBDF928, 0x3B1FF8,
  0x97FFDE, 0x05980F, 0xEF2F11, 0x8B5A0A, 0x6D1F6D, 0x367ECF,
  0x27CB09, 0xB74F46, 0x3F669E, 0x5FEA2D, 0x7527BA, 0xC7EBE5,
  0xF17B3D, 0x0739F7, 0x8A5292, 0xEA6BFB, 0x5FB11F, 0x8D5D08,
  0x560330, 0x46FC7B, 0x6BABF0, 0xCFBC20, 0x9AF436, 0x1DA9E3,
  0x91615E, 0xE61B08, 0x659985, 0x5F14A0, 0x68408D, 0xFFD880,
  0x4D7327, 0x310606, 0x1556CA, 0x73A8C9, 0x60E27B, 0xC08C6B,
};
__device__ static const double ZM_PIO2[8] = {
  0x1.921fb40000000p+0,  0x1.4442d00000000p-24, 0x1.8469880000000p-48,
  0x1.8cc5160000000p-72, 0x1.01b8380000000p-96, 0x1.a252040000000p-120,
  0x1.3822280000000p-145, 0x1.9f31d00000000p-169,
};

// Payne-Hanek reduction (musl __rem_pio2_large with nx = 1, prec = 0).  Kept out of line:
// it is the cold path and its work arrays live in scratch.  Result by value and declared `pure` (it reads its arguments
// and the constant tables, writes nothing the caller can see): a call whose result is unused can then be deleted -- the
// frame-range launches of generated kernels rely on that to drop a sine that only feeds the discarded output
// (script_rt.hip.h zs_frame_loop); through an out-pointer the call kept every such sine alive.
struct ZRemLarge { int n; double y; };
__device__ __noinline__ __attribute__((pure)) static ZRemLarge zrem_pio2_large1(double x0, int e0) {
    int32_t jz, jv, carry, n, iq[20], i, j, k, q0, ih;
    const int32_t jk = 3;
    double z, fw, f[20], fq[20], q[20];
    jv = (e0 - 3) / 24; if (jv < 0) jv = 0;
    q0 = e0 - 24 * (jv + 1);
    for (i = 0; i <= jk; i++) f[i] = (double)ZM_IPIO2[jv + i];
    for (i = 0; i <= jk; i++) q[i] = x0 * f[i];
    jz = jk;
    for (;;) {
        for (i = 0, j = jz, z = q[jz]; j > 0; i++, j--) {
            fw = (double)(int32_t)(0x1p-24 * z);
            iq[i] = (int32_t)(z - 0x1p24 * fw);
            z = q[j - 1] + fw;
        }
        z = ldexp(z, q0);
        z -= 8.0 * floor(z * 0.125);
        n = (int32_t)z;
        z -= (double)n;
        ih = 0;
        if (q0 > 0) {
            i = iq[jz - 1] >> (24 - q0); n += i;
            iq[jz - 1] -= i << (24 - q0);
            ih = iq[jz - 1] >> (23 - q0);
        } else if (q0 == 0) ih = iq[jz - 1] >> 23;
        else if (z >= 0.5) ih = 2;
        if (ih > 0) {
            n += 1; carry = 0;
            for (i = 0; i < jz; i++) {
                j = iq[i];
                if (carry == 0) {
                    if (j != 0) { carry = 1; iq[i] = 0x1000000 - j; }
                } else iq[i] = 0xffffff - j;
            }
            if (q0 == 1) iq[jz - 1] &= 0x7fffff;
            else if (q0 == 2) iq[jz - 1] &= 0x3fffff;
            if (ih == 2) {
                z = 1.0 - z;
                if (carry != 0) z -= ldexp(1.0, q0);
            }
        }
        if (z == 0.0) {
            j = 0;
            for (i = jz - 1; i >= jk; i--) j |= iq[i];
            if (j == 0) {
                for (k = 1; iq[jk - k] == 0; k++) {}
                for (i = jz + 1; i <= jz + k; i++) {
                    f[i] = (double)ZM_IPIO2[jv + i];
                    q[i] = x0 * f[i];
                }
                jz += k;
                continue;
            }
        }
        break;
    }
    if (z == 0.0) {
        jz -= 1; q0 -= 24;
        while (iq[jz] == 0) { jz--; q0 -= 24; }
    } else {
        z = ldexp(z, -q0);
        if (z >= 0x1p24) {
            fw = (double)(int32_t)(0x1p-24 * z);
            iq[jz] = (int32_t)(z - 0x1p24 * fw);
            jz += 1; q0 += 24;
            iq[jz] = (int32_t)fw;
        } else iq[jz] = (int32_t)z;
    }
    fw = ldexp(1.0, q0);
    for (i = jz; i >= 0; i--) { q[i] = fw * (double)iq[i]; fw *= 0x1p-24; }
    for (i = jz; i >= 0; i--) {
        for (fw = 0.0, k = 0; k <= jk && k <= jz - i; k++) fw += ZM_PIO2[k] * q[i + k];
        fq[jz - i] = fw;
    }
    fw = 0.0;
    for (i = jz; i >= 0; i--) fw += fq[i];
    return ZRemLarge{n & 7, ih == 0 ? fw : -fw};
}

ZD int zrem_pio2f(float x, double *y) {
    const double toint = 1.5 / 2.220446049250313e-16, pio4 = 0x1.921fb6p-1,
                 invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079631090164184570e+00,
                 pio2_1t = 1.58932547735281966916e-08;
    uint32_t ui = zf2u(x), ix = ui & 0x7fffffff;
    if (ix < 0x4dc90fdb) {
        double fn = (double)x * invpio2 + toint - toint;
        int n = (int32_t)fn;
        *y = x - fn * pio2_1 - fn * pio2_1t;
        if (*y < -pio4) { n--; fn--; *y = x - fn * pio2_1 - fn * pio2_1t; }
        else if (*y > pio4) { n++; fn++; *y = x - fn * pio2_1 - fn * pio2_1t; }
        return n;
    }
    if (ix >= 0x7f800000) { *y = x - x; return 0; }
    int sign = ui >> 31;
    int e0 = (int)(ix >> 23) - (0x7f + 23);
    const ZRemLarge big = zrem_pio2_large1((double)zu2f(ix - ((uint32_t)e0 << 23)), e0);
    if (sign) { *y = -big.y; return -big.n; }
    *y = big.y;
    return big.n;
}

// musl sinf / cosf (what Zig's std.math.sin / cos are ported from): reduce x to y in [-pi/4, pi/4] with
// x = y + n*pi/2 (in double), then one of two polynomial kernels picked by n & 3.  musl spells the
// reduction out as five magnitude ranges times two signs for |x| <= 9pi/4 (x -+ k*M_PI_2, one constant), a
// two-constant form up to 2^28*pi/2 (fn = x*invpio2 + 1.5*2^52 - 1.5*2^52, y = x - fn*pio2_1 - fn*pio2_1t, two
// correction branches) and a general routine above.  A wave's lanes sit in all of those leaves at once (oscillator
// phases are spread), so branches cost every lane every leaf.  Round 2 folded the leaves into straight-line code
// performing, per lane, exactly the operations of that lane's leaf (both reductions, a select); round 3 replaced
// "the same operations" by "the same bits for every argument", established by exhaustion (the list inside the
// function): ONE reduction for every |x| below the Payne-Hanek range, the large-argument routine (and inf / nan)
// behind one wave-uniform test.  Then ONE evaluation of each kernel on y (z = y*y and w = z*z shared) and
// selects: the kernels are exactly odd / even in floating point, so sindf(-y) = -sindf(y) and every negation is a
// sign-bit flip of the result.
template <bool MAYBE_LARGE = true>
ZD int zreduce_pio2f(float x, uint32_t ix, double &y) {
    const double invpio2 = 6.36619772367581382433e-01,
                 pio2_1 = 1.57079631090164184570e+00, pio2_1t = 1.58932547735281966916e-08,
                 toint = 1.5 / 2.220446049250313e-16;
    const double xd = (double)x;
    // Round 3: every step below is the cheapest form that gives the SAME BITS as musl's for every f32 argument, decided by
    // exhaustion (tools/ubench/sin_exhaustive.hip, all 2^32 patterns, sinf and cosf; profiles/r03/sin_exhaustive.txt):
    //  * fn by musl's own magic-number rounding, x*invpio2 + 1.5*2^52 - 1.5*2^52, with the product fused into the add; the
    //    integer n is then the low mantissa word of the sum (two's complement): no v_rndne_f64, no v_cvt_i32_f64;
    //  * the products of the reduction fused into its subtractions;
    //  * no "|y| > pi/4" correction test: musl says it "matters with directed rounding" -- under round-to-nearest it changes
    //    no result; what is left of the rare path is |x| >= 2^28*pi/2 (Payne-Hanek), and inf / nan, whose y = x - x runs
    //    through the kernels to the same NaN musl's early return gives.
    //  * ONE leaf: musl reduces |x| <= 9pi/4 as x - k*pi/2 with a single constant and everything above with the two-constant
    //    form; the two-constant form applied to EVERY argument gives the same sinf and cosf bits for all 2^32 arguments but
    //    one magnitude -- |x| = 0x40406406 (3.0061...), where sinf comes out one ulp high -- which zsinf corrects by name
    //    (kZSinOneLeafOdd): no second reduction, no select.
    const double fnm = __builtin_fma(xd, invpio2, toint);
    const double fn = fnm - toint;
    int n = (int)(uint32_t)__double_as_longlong(fnm);
    y = __builtin_fma(-fn, pio2_1t, __builtin_fma(-fn, pio2_1, xd));                       // xd - fn * pio2_1 - fn * pio2_1t
    // (the compare as a wave mask: the ballot of a combined per-lane bool costs a 0/1 select and a compare more)
    // MAYBE_LARGE = false: the caller has established (wave-wide, for a whole chunk of frames: kZSinNoLargeBelow) that no
    // argument reaches the rare path -- without its branch the sines of an unrolled chunk are one basic block
    if (MAYBE_LARGE && __builtin_expect(__builtin_amdgcn_ballot_w64(ix > 0x4dc90fdau) != 0, 0)) {
        if (ix > 0x4dc90fdau) n = zrem_pio2f(x, &y);                  // finite |x| >= 2^28*pi/2, inf, nan
    }
    return n;
}

// The two polynomial kernels of musl's sinf / cosf on the reduced argument, both at once (z and w shared):
//   __sindf(y) = (y + s*(S1 + z*S2)) + s*w*(S3 + z*S4),  s = z*y          __cosdf(y) = ((1 + z*C0) + w*C1) + (w*z)*(C2 + z*C3)
// musl rounds every product and every sum (13 + 5 f64 operations with the reduction's); here every multiply-add pair is ONE
// fused operation (nine fewer f64 instructions per sine).  A fused multiply-add skips one f64 rounding, which moves
// the f32 result only if the f64 value then crosses a rounding boundary of the final conversion; whether that ever happens
// is decided by exhaustion, not by argument: tools/ubench/sin_exhaustive.hip runs ALL 2^32 f32 bit patterns through musl's
// operation order and through this form -- sinf and cosf, every leaf of the reduction -- and finds 0 differing results
// (profiles/r03/sin_exhaustive.txt; each of the fusions alone, and all together).  tests/test_gpu_math.py repeats a 2^28-argument stratified subset against the
// oracle, which keeps musl's order.
ZD void zsincos_kernels(double y, float &sv, float &cv) {
    const double S1 = -0x15555554cbac77.0p-55, S2 = 0x111110896efbb2.0p-59, S3 = -0x1a00f9e2cae774.0p-65, S4 = 0x16cd878c3b46a7.0p-71;
    const double C0 = -0x1ffffffd0c5e81.0p-54, C1 = 0x155553e1053a42.0p-57, C2 = -0x16c087e80f1e27.0p-62, C3 = 0x199342e0ee5069.0p-68;
    const double z = y * y, w = z * z, s = z * y;
    sv = (float)__builtin_fma(s * w, __builtin_fma(z, S4, S3), __builtin_fma(s, __builtin_fma(z, S2, S1), y));
    cv = (float)__builtin_fma(w * z, __builtin_fma(z, C3, C2), __builtin_fma(w, C1, __builtin_fma(z, C0, 1.0)));
}

constexpr uint32_t kZSinOneLeafOdd = 0x40406406u;            // see zreduce_pio2f ("ONE leaf"); found and checked by tools/ubench/sin_exhaustive.hip
// |x| below this (NaN excluded by the comparison) never takes the rare path: zsinf<false> / zcosf<false> are then exact
constexpr float kZSinNoLargeBelow = 4.0e8f;                    // < 2^28 * pi/2 = 4.2166e8 (0x4dc90fdb)
template <bool MAYBE_LARGE = true>
ZD float zsinf(float x) {
    const uint32_t ux = zf2u(x), ix = ux & 0x7fffffff;
    double y;
    const int n = zreduce_pio2f<MAYBE_LARGE>(x, ix, y);
    float sv, cv;
    zsincos_kernels(y, sv, cv);
    // n & 3: 0 sindf(y) | 1 cosdf(y) | 2 sindf(-y) = -sindf(y) | 3 -cosdf(y)
    float r = zu2f(zf2u((n & 1) ? cv : sv) ^ ((uint32_t)(n & 2) << 30));
    r = zu2f(zf2u(r) - (uint32_t)(ix == kZSinOneLeafOdd));    // the one magnitude where the single-leaf reduction is an ulp high
    if (ix < 0x39800000) r = x;                               // |x| < 2^-12 (needed for -0.0 alone: the kernel gives +0.0)
    return r;                                                 // (inf, nan: y = x - x from the rare path, NaN through the kernel)
}

// ZH_PAINT_TOLERANT (opt-in; include/zang_hip.h): sin(x) in f32 alone -- k = rint(x / pi), r = x - k pi by two fused steps
// (pi as a float pair: the first step is exact, the pair is good for |k| < 2^20), the odd Taylor polynomial to r^11 on
// [-pi/2, pi/2] (last term (pi/2)^13 / 13! = 5.7e-8), the sign from k's low bit: 17 f32 instructions for musl's 34 (15 of them
// f64).  Within 2.4e-7 of musl's sinf of the SAME argument for |x| < 2^20 (tests/test_gpu_tolerant.py sweeps it); arguments
// beyond, inf and nan take the exact routine (a wave-uniform branch).  What a SineOsc feeds it is the reference's own rounded
// argument ((t + phase) * pi) * 2, so the argument's rounding -- the larger effect, 3e-5 at x = 600 -- is reproduced, not skipped.
ZD float zsinf_tol(float x) {
    const float inv_pi = 0.31830988618379067154f, pi_hi = 3.14159274101257324219f, pi_lo = -8.74227765734758577e-8f;
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(__builtin_fabsf(x) < 1048576.0f)) != 0, 0)) {
        if (!(__builtin_fabsf(x) < 1048576.0f)) return zsinf<true>(x);
    }
    const float kf = __builtin_rintf(x * inv_pi);
    float r = __builtin_fmaf(-kf, pi_hi, x);
    r = __builtin_fmaf(-kf, pi_lo, r);
    const float z = r * r;
    float p = -2.50521083854417187751e-8f;                            // -1/11!
    p = __builtin_fmaf(p, z, 2.75573192239858906526e-6f);             //  1/9!
    p = __builtin_fmaf(p, z, -1.98412698412698412698e-4f);            // -1/7!
    p = __builtin_fmaf(p, z, 8.33333333333333333333e-3f);             //  1/5!
    p = __builtin_fmaf(p, z, -1.66666666666666666667e-1f);            // -1/3!
    const float sv = __builtin_fmaf(r * z, p, r);
    return zu2f(zf2u(sv) ^ ((uint32_t)(int)kf << 31));
}

template <bool MAYBE_LARGE = true>
ZD float zcosf(float x) {
    const uint32_t ux = zf2u(x), ix = ux & 0x7fffffff;
    double y;
    const int n = zreduce_pio2f<MAYBE_LARGE>(x, ix, y);
    float sv, cv;
    zsincos_kernels(y, sv, cv);
    // n & 3: 0 cosdf(y) | 1 sindf(-y) = -sindf(y) | 2 -cosdf(y) | 3 sindf(y)
    float r = zu2f(zf2u((n & 1) ? sv : cv) ^ ((uint32_t)((n + 1) & 2) << 30));
    return r;                                                 // (|x| < 2^-12: the kernel rounds to 1.0f by itself; inf, nan as in zsinf)
}

// ---- atanf (f32 arithmetic throughout) ------------------------------------------------
// musl atanf: five magnitude ranges, four of them reducing x by a different quotient before one shared
// polynomial.  Distortion's overdrive calls it per sample on arbitrary signal values, so a wave's lanes sit in all
// ranges at once; as branches every range's divide ran for every wave.  Round 2 folded it like sinf: each range's numerator
// and denominator by selects, ONE divide, then the shared polynomial and selects for the tails.
// Round 4, decided by exhaustion (tools/ubench/atan_exhaustive.hip, profiles/r04/atan_exhaustive.txt: all 2^32 arguments
// against musl's own branchy order, 0 differing results for every step taken):
//   * UNIFIED: the three quotient ranges as num = |x| - c, den = 1 + c |x| with c = 1/2, 1, 3/2 -- the first is musl's
//     (2x - 1) / (2 + x) scaled by an exact 1/2 (same quotient bits), c |x| is exact for 1/2 and 1 and musl's own rounded product
//     for 3/2 -- so one select picks c and two more the -1 / |x| range: 4 selects instead of 6;
//   * RCP1: the divide as v_rcp_f32 and ONE residual correction of the quotient, fma(fma(-den, q, num), r, q): correctly
//     rounded for every operand pair atanf forms (denominators in [1, 2^26]: no scaling, no fix-up) -- 4 instructions for ~10.
// Also bit-identical but not taken: one coefficient row per range instead of selects (fast in the sweep's loop, but the rows
// become per-lane loads from a constant table: Distortion's overdrive at 131,072 voices went 232 -> 465 us).  Rejected by the
// sweep: each of the three polynomial fusions (34 / 2,876 / 256 differing arguments), dropping the |x| < 2^-12 early return (-0.0).
ZD float zatanf(float x) {
    const float aT0 = 3.3333328366e-01f, aT1 = -1.9999158382e-01f, aT2 = 1.4253635705e-01f,
                aT3 = -1.0648017377e-01f, aT4 = 6.1687607318e-02f;
    const uint32_t ux = zf2u(x), ix = ux & 0x7fffffff;
    const bool sign = (ux >> 31) != 0;
    const float ax = fabsf(x);
    const bool r0 = ix < 0x3f300000, r1 = ix < 0x3f980000, r2 = ix < 0x401c0000;   // |x| < 11/16, < 19/16, < 39/16
    // id 0: (x-1/2)/(1+x/2)   id 1: (x-1)/(1+x)   id 2: (x-3/2)/(1+3x/2)   id 3: -1/x
    const float c = r0 ? 0.5f : (r1 ? 1.0f : 1.5f);
    const float num = r2 ? ax - c : -1.0f;
    const float den = r2 ? 1.0f + c * ax : ax;
    const float hi = r1 ? (r0 ? 4.6364760399e-01f : 7.8539812565e-01f) : (r2 ? 9.8279368877e-01f : 1.5707962513e+00f);
    const float lo = r1 ? (r0 ? 5.0121582440e-09f : 3.7748947079e-08f) : (r2 ? 3.4473217170e-08f : 7.5497894159e-08f);
    const bool direct = ix < 0x3ee00000;                              // |x| < 7/16: id = -1, x itself (signed)
    const float rc = __builtin_amdgcn_rcpf(den);
    const float q = num * rc;
    const float xr = direct ? x : __builtin_fmaf(__builtin_fmaf(-den, q, num), rc, q);
    const float z = xr * xr;
    const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * aT4));
    const float s2 = w * (aT1 + w * aT3);
    const float t = xr * (s1 + s2);
    const float zz = hi - ((t - lo) - xr);
    float r = direct ? xr - t : (sign ? -zz : zz);
    if (ix < 0x39800000) r = x;                                       // |x| < 2^-12
    if (ix >= 0x4c800000) {                                           // |x| >= 2^26, inf, nan
        const float big = 1.5707962513e+00f + 0x1p-120f;
        r = (x != x) ? x : (sign ? -big : big);
    }
    return r;
}

// ---- logf / expf / powf (Distortion.zig:41 calls pow(f32, 2.0, y) once per paint) -------
ZD float zlogf(float x) {
    const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f, Lg1 = 0xaaaaaa.0p-24f,
                Lg2 = 0xccce13.0p-25f, Lg3 = 0x91e9ee.0p-25f, Lg4 = 0xf89e26.0p-26f;
    uint32_t ix = zf2u(x);
    int k = 0;
    if (ix < 0x00800000 || ix >> 31) {
        if (ix << 1 == 0) return -__builtin_inff();
        if (ix >> 31) return __builtin_nanf("");
        k -= 25; x *= 0x1p25f; ix = zf2u(x);
    } else if (ix >= 0x7f800000) return x;
    else if (ix == 0x3f800000) return 0;
    ix += 0x3f800000 - 0x3f3504f3;
    k += (int)(ix >> 23) - 0x7f;
    ix = (ix & 0x007fffff) + 0x3f3504f3;
    x = zu2f(ix);
    float f = x - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float w = z * z;
    float t1 = w * (Lg2 + w * Lg4);
    float t2 = z * (Lg1 + w * Lg3);
    float R = t2 + t1;
    float hfsq = 0.5f * f * f;
    float dk = (float)k;
    return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

ZD float zexpf(float x) {
    const float ln2hi = 6.9314575195e-1f, ln2lo = 1.4286067653e-6f, invln2 = 1.4426950216e+0f,
                P1 = 1.6666625440e-1f, P2 = -2.7667332906e-3f;
    uint32_t hx = zf2u(x);
    int sign = hx >> 31, k;
    float hi, lo;
    hx &= 0x7fffffff;
    if (x != x) return x;
    if (hx >= 0x42aeac50) {
        if (hx > 0x7f800000) return x;
        if (hx >= 0x42b17218 && !sign) return x * 0x1p127f;
        if (sign && hx >= 0x42cff1b5) return 0;
    }
    if (hx > 0x3eb17218) {
        if (hx > 0x3f851592) k = (int)(invln2 * x + (sign ? -0.5f : 0.5f));
        else k = 1 - sign - sign;
        float fk = (float)k;
        hi = x - fk * ln2hi;
        lo = fk * ln2lo;
        x = hi - lo;
    } else if (hx > 0x39000000) {
        k = 0; hi = x; lo = 0;
    } else return 1 + x;
    float xx = x * x;
    float c = x - xx * (P1 + xx * P2);
    float y = 1 + (x * c / (2 - c) - lo + hi);
    if (k == 0) return y;
    return ldexpf(y, k);
}

// std.math.pow(f32, x, y) restricted to finite x > 0 (the reference's only call is x = 2).
ZD float zpowf_pos(float x, float y) {
    if (y == 0 || x == 1) return 1;
    if (y != y) return __builtin_nanf("");
    if (y == 1) return x;
    if (__builtin_isinf(y)) {
        if ((x < 1) == (y > 0)) return 0;
        return __builtin_inff();
    }
    if (y == 0.5f) return sqrtf(x);
    if (y == -0.5f) return 1 / sqrtf(x);
    float ay = fabsf(y);
    float yi = truncf(ay);
    float yf = ay - yi;
    if (yi >= 2147483648.0f) return zexpf(y * zlogf(x));
    float a1 = 1.0f;
    int ae = 0;
    if (yf != 0) {
        if (yf > 0.5f) { yf -= 1; yi += 1; }
        a1 = zexpf(yf * zlogf(x));
    }
    int xe;
    float x1 = frexpf(x, &xe);
    for (int32_t i = (int32_t)yi; i != 0; i >>= 1) {
        if (xe < -(1 << 9) || (1 << 9) < xe) { ae += xe; break; }
        if (i & 1) { a1 *= x1; ae += xe; }
        x1 *= x1;
        xe <<= 1;
        if (x1 < 0.5f) { x1 += x1; xe -= 1; }
    }
    if (y < 0) { a1 = 1 / a1; ae = -ae; }
    return ldexpf(a1, ae);
}

// std.math.pow(f32, x, y) for any arguments (zangscript's `pow`, codegen_zig.zig:185): the special cases
// of the Go math.Pow port in front of the same core.
ZD float zpowf(float x, float y) {
    if (y == 0 || x == 1) return 1;
    if (x != x || y != y) return __builtin_nanf("");
    if (y == 1) return x;
    if (x == 0) {
        bool y_odd_int = false;
        if (fabsf(y) < 16777216.0f) { const float yi = truncf(y); y_odd_int = (yi == y) && ((int32_t)yi & 1); }
        if (y < 0) return y_odd_int ? copysignf(__builtin_inff(), x) : __builtin_inff();
        return y_odd_int ? x : 0.0f;
    }
    if (__builtin_isinf(y)) {
        if (x == -1) return 1;
        if ((fabsf(x) < 1) == (y > 0)) return 0;
        return __builtin_inff();
    }
    if (__builtin_isinf(x)) {
        if (x < 0) {                                   // pow(1 / x, -y) with 1 / x == -0
            const float ny = -y;
            bool odd = false;
            if (fabsf(ny) < 16777216.0f) { const float yi = truncf(ny); odd = (yi == ny) && ((int32_t)yi & 1); }
            if (ny < 0) return odd ? -__builtin_inff() : __builtin_inff();
            return odd ? -0.0f : 0.0f;
        }
        return y < 0 ? 0.0f : __builtin_inff();
    }
    if (y == 0.5f) return sqrtf(x);
    if (y == -0.5f) return 1 / sqrtf(x);
    float ay = fabsf(y);
    float yi = truncf(ay);
    float yf = ay - yi;
    if (yf != 0 && x < 0) return __builtin_nanf("");
    if (yi >= 2147483648.0f) return zexpf(y * zlogf(x));
    float a1 = 1.0f;
    int ae = 0;
    if (yf != 0) {
        if (yf > 0.5f) { yf -= 1; yi += 1; }
        a1 = zexpf(yf * zlogf(x));
    }
    int xe;
    float x1 = frexpf(x, &xe);
    for (int32_t i = (int32_t)yi; i != 0; i >>= 1) {
        if (xe < -(1 << 9) || (1 << 9) < xe) { ae += xe; break; }
        if (i & 1) { a1 *= x1; ae += xe; }
        x1 *= x1;
        xe <<= 1;
        if (x1 < 0.5f) { x1 += x1; xe -= 1; }
    }
    if (y < 0) { a1 = 1 / a1; ae = -ae; }
    return ldexpf(a1, ae);
}

// ---- xoshiro256++ and Random.float(f32) (Noise.zig:22,29,51,58) -------------------------
struct ZXoshiro { uint64_t s0, s1, s2, s3; };

ZD uint64_t zrotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

ZD void zxoshiro_seed(ZXoshiro &r, uint64_t seed) {
    uint64_t sm = seed, o[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        sm += 0x9e3779b97f4a7c15ull;
        uint64_t z = sm;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        o[i] = z ^ (z >> 31);
    }
    r.s0 = o[0]; r.s1 = o[1]; r.s2 = o[2]; r.s3 = o[3];
}
ZD uint64_t zxoshiro_next(ZXoshiro &r) {
    uint64_t result = zrotl64(r.s0 + r.s3, 23) + r.s0;
    uint64_t t = r.s1 << 17;
    r.s2 ^= r.s0; r.s3 ^= r.s1; r.s1 ^= r.s2; r.s0 ^= r.s3;
    r.s2 ^= t;
    r.s3 = zrotl64(r.s3, 45);
    return result;
}
ZD float zrandom_float32(ZXoshiro &r) {
    const uint64_t rnd = zxoshiro_next(r);
    const uint32_t hi = (uint32_t)(rnd >> 32);
    uint32_t lz;
    asm("v_ffbh_u32 %0, %1" : "=v"(lz) : "v"(hi));                   // the whole answer whenever hi != 0 (any value when hi == 0)
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(hi == 0u) != 0, 0)) {   // wave-uniform test: one compare + one scalar branch
        if (hi == 0u) {                                              // probability 2^-32 per sample: the general form
            lz = rnd ? (uint32_t)__clzll((long long)rnd) : 64u;
            if (lz >= 41) {                                          // probability 2^-41 per sample
                uint64_t r2 = zxoshiro_next(r);
                lz = 41 + (r2 ? (uint32_t)__clzll((long long)r2) : 64u);
                if (lz == 41 + 64) lz += (uint32_t)__clz((int)((uint32_t)zxoshiro_next(r) | 0x7FFu));
            }
        }
    }
    return zu2f(((126u - lz) << 23) | ((uint32_t)rnd & 0x7FFFFFu));
}

// ---- v2 oscillator helpers (PulseOsc.zig:12-26, TriSawOsc.zig:8-26) ----------------------
ZD float zclamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }
ZD float zutof23(uint32_t x) { return zu2f((x >> 9) | 0x3f800000u) - 1.0f; }
ZD uint32_t zftou32(float v) { return zf32_to_u32(v * 4294967296.0f * 0.99995f); }

// Filter.cutoffFromFrequency (Filter.zig:20-23)
ZD float zcutoff_from_frequency(float frequency, float sample_rate) {
    const float v = 2.0f * (1.0f - zcosf(3.14159265358979323846f * frequency / sample_rate));
    return sqrtf(zclampf(v, 0.0f, 1.0f));
}
